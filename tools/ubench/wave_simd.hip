// Which SIMD do the four waves of a 256-thread workgroup land on?  (HW_ID bits 5:4 = SIMD, 11:8 = CU, 15:13 = SE on gfx9.)
// hipcc --offload-arch=gfx950 -O2 tools/ubench/wave_simd.hip -o /tmp/wave_simd && /tmp/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
    if ((threadIdx.x & 63u) == 0u) out[blockIdx.x * 4u + (threadIdx.x >> 6)] = hw;
    // stay resident for a while so that several workgroups share a CU
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000ull) {}
}
int main() {
    const unsigned groups = 256 * 6;
    unsigned* d; hipMalloc((void**)&d, groups * 16);
    hipLaunchKernelGGL(probe, dim3(groups), dim3(256), 0, 0, d);
    std::vector<unsigned> h(groups * 4);
    hipMemcpy(h.data(), d, groups * 16, hipMemcpyDeviceToHost);
    unsigned hist[4][4] = {}, distinct = 0;
    for (unsigned g = 0; g < groups; ++g) {
        unsigned mask = 0;
        for (unsigned w = 0; w < 4; ++w) { const unsigned simd = (h[g * 4 + w] >> 4) & 3u; hist[w][simd]++; mask |= 1u << simd; }
        distinct += mask == 15u;
    }
    for (unsigned w = 0; w < 4; ++w) printf("wave %u: SIMD0 %u SIMD1 %u SIMD2 %u SIMD3 %u\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("workgroups with one wave on each SIMD: %u of %u\n", distinct, groups);
    for (unsigned g = 0; g < 6; ++g) printf("wg %u: hw_id %08x %08x %08x %08x\n", g, h[g * 4], h[g * 4 + 1], h[g * 4 + 2], h[g * 4 + 3]);
    return 0;
}
