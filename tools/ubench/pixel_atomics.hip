// Throughput of the rasteriser's per-pixel visibility update on gfx950: every lane one 64-bit atomicMax on its own word,
// a wave covering an 8x8 pixel block of a 3840-wide buffer (8 rows x 64 B), against plain 8-byte stores and 32-bit
// atomics of the same shape.   hipcc --offload-arch=gfx950 -O2 pixel_atomics.hip -o pixel_atomics && ./pixel_atomics
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int BW = 8>   // BW x (64 / BW) pixels per wave
__global__ __launch_bounds__(256) void touch(unsigned long long* vis, unsigned width, unsigned blocks_x, unsigned nblocks, unsigned passes) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6), waves = gridDim.x * 4u;
    constexpr unsigned BH = 64 / BW;
    for (unsigned p = 0; p < passes; ++p)
        for (unsigned b = wave; b < nblocks; b += waves) {
            const unsigned bx = b % blocks_x, by = b / blocks_x;
            const size_t pix = (size_t)(by * BH + lane / BW) * width + bx * BW + (lane % BW);
            const unsigned long long v = ((unsigned long long)(p * 977u + lane) << 32) | b;
            if (MODE == 0) atomicMax(&vis[pix], v);
            else if (MODE == 1) vis[pix] = v;
            else if (MODE == 2) atomicMax(reinterpret_cast<unsigned*>(&vis[pix]), (unsigned)(v >> 32));
            else if (MODE == 3) { if (vis[pix] < v) atomicMax(&vis[pix], v); }
        }
}

int main() {
    const unsigned w = 3840, h = 2160, bxn = w / 8, nblocks = bxn * (h / 8), passes = 4;
    unsigned long long* vis;
    hipMalloc(&vis, (size_t)w * h * 8);
    const char* names[] = {"64-bit atomicMax", "8-byte store", "32-bit atomicMax", "load + compare, atomicMax if greater"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(vis, 0, (size_t)w * h * 8);
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a);
            if (mode == 0) touch<0><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            if (mode == 1) touch<1><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            if (mode == 2) touch<2><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            if (mode == 3) touch<3><<<2048, 256>>>(vis, w, bxn, nblocks, passes);
            hipEventRecord(b);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2)
                printf("%-40s %u passes over %ux%u: %.1f us per pass, %.1f G lane-ops/s\n", names[mode], passes, w, h, ms * 1e3 / passes,
                       (double)w * h * passes / (ms * 1e-3) / 1e9);
        }
    }
    // the shape of the wave's footprint: 8x8, 16x4, 32x2, 64x1 pixels (64 B, 128 B, 256 B, 512 B contiguous per row)
    for (int shape = 0; shape < 4; ++shape) {
        const unsigned bw = 8u << shape, bh = 64 / bw, bxs = w / bw, nb = bxs * (h / bh);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(vis, 0, (size_t)w * h * 8);
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a);
            if (shape == 0) touch<0, 8><<<2048, 256>>>(vis, w, bxs, nb, passes);
            if (shape == 1) touch<0, 16><<<2048, 256>>>(vis, w, bxs, nb, passes);
            if (shape == 2) touch<0, 32><<<2048, 256>>>(vis, w, bxs, nb, passes);
            if (shape == 3) touch<0, 64><<<2048, 256>>>(vis, w, bxs, nb, passes);
            hipEventRecord(b);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("64-bit atomicMax, wave footprint %2ux%u: %.1f us per pass\n", bw, bh, ms * 1e3 / passes);
        }
    }
    return 0;
}
