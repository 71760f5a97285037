// Does s_atomic_add (scalar-unit atomic, result in an SGPR, counted on lgkmcnt) work on gfx950, and how fast is a
// same-address stream of them?  Every wave takes `per_wave` tickets from one of `ncounters` counters; the tickets must
// be a permutation.   hipcc --offload-arch=gfx950 -O2 scalar_atomic.hip -o scalar_atomic && ./scalar_atomic
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void take(unsigned* counters, unsigned ncounters, unsigned per_wave, unsigned* tickets, int vector_path) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
    unsigned* c = counters + (wave % ncounters) * 64;
    for (unsigned i = 0; i < per_wave; ++i) {
        unsigned t;
        if (vector_path) {
            unsigned v = 0;
            if ((threadIdx.x & 63) == 0) v = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = __builtin_amdgcn_readfirstlane(v);
        } else {
            t = 1;
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(c) : "memory");
        }
        if ((threadIdx.x & 63) == 0) tickets[(size_t)wave * per_wave + i] = t * ncounters + wave % ncounters;
    }
}

int main() {
    const unsigned blocks = 2048, waves = blocks * 4, per_wave = 16;
    for (int vector_path = 0; vector_path < 2; ++vector_path)
        for (unsigned nc : {1u, 8u, 64u, 512u}) {
            unsigned *counters, *tickets;
            hipMalloc(&counters, nc * 256);
            hipMemset(counters, 0, nc * 256);
            hipMalloc(&tickets, (size_t)waves * per_wave * 4);
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a);
            take<<<blocks, 256>>>(counters, nc, per_wave, tickets, vector_path);
            hipEventRecord(b);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            float ms;
            hipEventElapsedTime(&ms, a, b);
            std::vector<unsigned> h((size_t)waves * per_wave);
            hipMemcpy(h.data(), tickets, h.size() * 4, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            bool ok = std::adjacent_find(h.begin(), h.end()) == h.end() && h.back() == h.size() - 1;
            printf("%s atomics, %3u counters: %zu tickets in %.1f us (%.1f ns per ticket overall) %s\n",
                   vector_path ? "vector" : "scalar", nc, h.size(), ms * 1e3, ms * 1e6 / h.size(), ok ? "permutation ok" : "TICKETS WRONG");
            hipFree(counters);
            hipFree(tickets);
        }
    return 0;
}
