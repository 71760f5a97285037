// Where do the rasteriser's 64-bit atomicMax operations execute, and what do they cost there?  gfx950 has one L2 per XCD
// (not coherent with the others): an agent-scope atomic cannot be resolved in an XCD's L2; a workgroup-scope one is — it
// is atomic among all waves of that XCD (they share the L2), which is enough when each screen region is touched by ONE
// XCD only.  Every lane one atomicMax on its own word of a 3840x2160 buffer, 8x8 pixel blocks, `passes` passes (so words
// are hit repeatedly), blocks dealt (a) round-robin over all waves, (b) by XCD: the workgroup on XCD x (hardware places
// workgroup b on XCD b % 8) takes blocks of row band x only.  The final buffer is checked against the expected maxima.
// hipcc --offload-arch=gfx950 -O2 atomic_scope.hip -o atomic_scope && ./atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE /* 0 agent, 1 workgroup, 2 plain store (no atomic) */, bool BY_XCD>
__global__ __launch_bounds__(256) void touch(unsigned long long* vis, unsigned width, unsigned blocks_x, unsigned nblocks, unsigned passes) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned xcd = blockIdx.x & 7u, per_xcd = nblocks / 8u;          // (nblocks is a multiple of 8 here)
    const unsigned slot = BY_XCD ? (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6) : blockIdx.x * 4u + (threadIdx.x >> 6);
    const unsigned step = BY_XCD ? (gridDim.x >> 3) * 4u : gridDim.x * 4u;
    const unsigned first = BY_XCD ? xcd * per_xcd : 0u, count = BY_XCD ? per_xcd : nblocks;
    for (unsigned p = 0; p < passes; ++p)
        for (unsigned i = slot; i < count; i += step) {
            const unsigned b = first + i;
            const unsigned bx = b % blocks_x, by = b / blocks_x;
            const size_t pix = (size_t)(by * 8u + lane / 8u) * width + bx * 8u + (lane % 8u);
            const unsigned long long v = ((unsigned long long)(((p * 2654435761u) ^ (b * 40503u) ^ lane) & 0xFFFFFu) << 32) | b;
            if (SCOPE == 0) __hip_atomic_fetch_max(&vis[pix], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (SCOPE == 1) __hip_atomic_fetch_max(&vis[pix], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else vis[pix] = v;
        }
}

int main() {
    const unsigned w = 3840, h = 2160, bxn = w / 8, nblocks = bxn * (h / 8), passes = 4;
    unsigned long long* vis;
    hipMalloc(&vis, (size_t)w * h * 8);
    std::vector<unsigned long long> want((size_t)w * h, 0), got((size_t)w * h);
    for (unsigned p = 0; p < passes; ++p)
        for (unsigned b = 0; b < nblocks; ++b)
            for (unsigned lane = 0; lane < 64; ++lane) {
                const unsigned bx = b % bxn, by = b / bxn;
                const size_t pix = (size_t)(by * 8u + lane / 8u) * w + bx * 8u + (lane % 8u);
                const unsigned long long v = ((unsigned long long)(((p * 2654435761u) ^ (b * 40503u) ^ lane) & 0xFFFFFu) << 32) | b;
                if (v > want[pix]) want[pix] = v;
            }
    const char* names[] = {"agent scope, blocks round-robin", "agent scope, blocks by XCD", "workgroup scope, blocks by XCD", "plain store, blocks by XCD",
                           "workgroup scope, blocks round-robin (NOT coherent: for the rate only)"};
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(vis, 0, (size_t)w * h * 8);
            hipDeviceSynchronize();
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a);
            if (mode == 0) touch<0, false><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            if (mode == 1) touch<0, true><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            if (mode == 2) touch<1, true><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            if (mode == 3) touch<2, true><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            if (mode == 4) touch<1, false><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            hipEventRecord(b);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        hipMemcpy(got.data(), vis, (size_t)w * h * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
        printf("%-72s %.1f us per pass, %.1f G lane-ops/s, %zu of %zu words differ from the expected maxima\n", names[mode],
               best * 1e3 / passes, (double)w * h * passes / (best * 1e-3) / 1e9, bad, got.size());
    }
    return 0;
}
