// Do atomics on DIFFERENT words of one 128-byte line queue up like atomics on one word?  8 192 waves take 16 tickets each from
// one of 32 counters (lane 0, atomic with return, the wave waits for each), the counters `stride` words apart: 1 = all 32 in
// one line, 32 = a line each, 64 = a line each with a line between.  One counter for reference.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/line_atomics.hip -o /tmp/line_atomics && /tmp/line_atomics
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void take(unsigned* counters, unsigned ncounters, unsigned stride, unsigned per_wave, unsigned* sink) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
    unsigned* c = counters + (size_t)(wave % ncounters) * stride;
    unsigned sum = 0;
    for (unsigned i = 0; i < per_wave; ++i) {
        unsigned v = 0;
        if ((threadIdx.x & 63) == 0) v = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += __builtin_amdgcn_readfirstlane(v);
    }
    if ((threadIdx.x & 63) == 0 && sum == 0xFFFFFFFFu) sink[0] = sum;
}

int main() {
    const unsigned blocks = 2048, per_wave = 16;
    unsigned *counters, *sink;
    hipMalloc(&counters, 32 * 64 * 4 + 256);
    hipMalloc(&sink, 4);
    struct { unsigned n, stride; const char* what; } cases[] = {
        {1, 1, "1 counter"}, {32, 1, "32 counters in one 128-byte line"}, {32, 32, "32 counters, a line each"}, {32, 64, "32 counters, 256 bytes apart"}};
    for (auto& k : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(counters, 0, 32 * 64 * 4 + 256);
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            take<<<blocks, 256>>>(counters, k.n, k.stride, per_wave, sink);
            hipEventRecord(b);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        const double n = (double)blocks * 4 * per_wave;
        printf("%-36s %8.1f us for %.0f tickets = %.1f ns per ticket (%.1f ns per ticket and counter)\n", k.what, best * 1e3, n, best * 1e6 / n, best * 1e6 / n * k.n);
    }
    return 0;
}
