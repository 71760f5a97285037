// tr_probe.h — measurement scaffolding for the shading kernels.  NOT part of the product: tr_kernels.h includes this file
// only when a profiling macro is defined, which only tools/build_variant.py does (build_ab/libtr_NAME.so;
// __graft_entry__.compile_library refuses such flags for transmission_renderer_amd/libtr_shade.so).
//   -DTR_ABLATION=1     fp.ablate (TR_ABLATE env, read per launch) switches phases off: bit0 no pyramid taps, bit1 no LUT,
//                       bit2 no sun, bit3 no punctual lights, bit5 pure streaming skeleton, bit6 no G-buffer traffic
//                       (synthetic inputs), bit7 no stores, bit8 refraction taps at the pixel's own place (a streaming pattern),
//                       bit9 every tap the same texels (no tap traffic, the same instructions)
//   -DTR_PROBE_MASK=n   the same phases compiled out (register-pressure probes, tools/kernel_stats.py)
//   -DTR_TIMING=1       every wave adds the cycles it waited for (0) the planes, (1) the cluster lists, (2) taps + LUT, (3)
//                       its loop time, (4) tiles into tr_timing_counters (tr_debug_read_timing; tools/gpu_timing_cold.py);
//                       slots 8 / 9: the time in a textured material's sampling front end / in the light loops; every wave of
//                       the frame recorder's opaque launch (-DTR_TIMING=2: of its transmissive launch) also logs its begin and
//                       end on the 100 MHz clock and its tiles (tr_shade_wave_log; TR_WAVE_LOG=path writes it as text)
#pragma once

#ifndef TR_ABLATION
#define TR_ABLATION 0
#endif
#ifndef TR_TIMING
#define TR_TIMING 0
#endif
#ifdef TR_PROBE_MASK
#define TR_ABLATE(L, bit) (((TR_PROBE_MASK) & (bit)) != 0)
#else
#define TR_ABLATE(L, bit) (TR_ABLATION && ((L)->fp.ablate & (bit)))
#endif

#if TR_TIMING
namespace tr {
__device__ unsigned long long tr_timing_counters[12][1024];   // spread over 1024 slots: same-address atomics serialise
__device__ unsigned long long tr_shade_wave_log[65536][4];     // per wave of the LAST opaque VIS launch: begin, end (100 MHz), tiles, covered tiles
__device__ __forceinline__ unsigned long long tr_now() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ void tr_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
struct tr_timer { unsigned long long wait[5]; };
}  // namespace tr
#define TR_PROBE_ARGS_DECL , tr_timer& timer
#define TR_PROBE_ARGS , timer
#define TR_PROBE_WAVE_BEGIN                                                    \
    tr_timer timer = {{0ull, 0ull, 0ull, 0ull, 0ull}};                                     \
    unsigned long long tiles_done = 0;                                         \
    const unsigned long long t_loop = tr_now();                                \
    const unsigned long long t_real = __builtin_amdgcn_s_memrealtime();   /* constant 100 MHz */
#if TR_TIMING == 3   /* only a wave's begin, end and tile count: no forced waits, the kernel's own schedule */
#define TR_PROBE_SINCE(name)
#define TR_PROBE_DRAIN
#define TR_PROBE_WAITED(slot, name)
#else
#define TR_PROBE_SINCE(name) const unsigned long long name = tr_now();
#define TR_PROBE_DRAIN tr_drain();
#define TR_PROBE_WAITED(slot, name) \
    tr_drain();                     \
    timer.wait[slot] += tr_now() - name;
#endif
#define TR_PROBE_TILE_DONE ++tiles_done;
#define TR_PROBE_WAVE_END                                                                                        \
    if (lane == 0) {                                                                                             \
        atomicAdd(&tr_timing_counters[0][blockIdx.x & 1023u], timer.wait[0]);                                    \
        atomicAdd(&tr_timing_counters[1][blockIdx.x & 1023u], timer.wait[1]);                                    \
        atomicAdd(&tr_timing_counters[2][blockIdx.x & 1023u], timer.wait[2]);                                    \
        atomicAdd(&tr_timing_counters[3][blockIdx.x & 1023u], tr_now() - t_loop);                                \
        atomicAdd(&tr_timing_counters[4][blockIdx.x & 1023u], tiles_done);                                       \
        atomicAdd(&tr_timing_counters[5][blockIdx.x & 1023u], 1ull);                                             \
        atomicMax(&tr_timing_counters[6][blockIdx.x & 1023u], tr_now() - t_loop);                                \
        atomicAdd(&tr_timing_counters[7][blockIdx.x & 1023u], __builtin_amdgcn_s_memrealtime() - t_real);        \
        atomicAdd(&tr_timing_counters[8][blockIdx.x & 1023u], timer.wait[3]);                                    \
        atomicAdd(&tr_timing_counters[9][blockIdx.x & 1023u], timer.wait[4]);                                    \
        if (VIS && TRANSMISSIVE == (TR_TIMING == 2) && blockIdx.x < 65536u) {   /* -DTR_TIMING=2: the transmissive VIS launch instead */                                                       \
            tr_shade_wave_log[blockIdx.x][0] = t_real;                                                           \
            tr_shade_wave_log[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();                                 \
            tr_shade_wave_log[blockIdx.x][2] = tiles_done;                                                       \
            tr_shade_wave_log[blockIdx.x][3] = timer.wait[0];                                                    \
        }                                                                                                        \
    }
#else
#define TR_PROBE_ARGS_DECL
#define TR_PROBE_ARGS
#define TR_PROBE_WAVE_BEGIN
#define TR_PROBE_SINCE(name)
#define TR_PROBE_DRAIN
#define TR_PROBE_WAITED(slot, name)
#define TR_PROBE_TILE_DONE
#define TR_PROBE_WAVE_END
#endif

// ---- host side (tr_shade.hip includes this block through TR_PROBE_HOST)
#ifdef TR_PROBE_HOST
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
// per launch: which phases are off (TR_ABLATE), and a coarser grid (TR_GRID_QUARTERS: quarters of the resident blocks)
#define TR_PROBE_FRAME_PARAMS(fp)                                                             \
    if (TR_ABLATION) {                                                                        \
        if (const char* e_ = std::getenv("TR_ABLATE")) (fp)->ablate = (uint32_t)std::atoi(e_); \
    }
#define TR_PROBE_GRID(bpx)                                                                                         \
    if (const char* e_ = std::getenv("TR_GRID_QUARTERS")) (bpx) = (bpx) / tr::kGridRounds * (uint32_t)std::atoi(e_) / 4u;
#if TR_TIMING
// (not declared in include/tr_shade.h) reads and clears the kernels' wait-cycle counters
extern "C" int32_t tr_debug_read_timing(unsigned long long out[12]) {
    static unsigned long long host[12][1024];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(tr::tr_timing_counters), sizeof(host)) != hipSuccess) return -1;
    for (int k = 0; k < 12; ++k) {
        out[k] = 0;
        for (int i = 0; i < 1024; ++i) out[k] += host[k][i];
    }
    unsigned long long longest = 0;   // the longest-lived wave
    for (int i = 0; i < 1024; ++i) longest = std::max(longest, host[6][i]);
    out[5] = out[5] | (longest << 32);
    if (std::getenv("TR_TIMING_DUMP")) {   // per XCD: longest-lived wave, mean loop time, tiles
        for (int x = 0; x < 8; ++x) {
            unsigned long long mx = 0, sum = 0, waves = 0, tiles = 0;
            for (int i = x; i < 1024; i += 8) {
                mx = std::max(mx, host[6][i]);
                sum += host[3][i];
                waves += host[5][i];
                tiles += host[4][i];
            }
            std::fprintf(stderr, "xcd %d: longest wave %llu, mean %llu, waves %llu, tiles %llu; per block slot (loop ticks/tiles):", x, mx, waves ? sum / waves : 0, waves, tiles);
            for (int i = x; i < 1024; i += 8 * 8) std::fprintf(stderr, " %llu/%llu", host[5][i] ? host[3][i] / host[5][i] : 0, host[5][i] ? host[4][i] / host[5][i] : 0);
            std::fprintf(stderr, "\n");
        }
    }
    if (const char* path = std::getenv("TR_WAVE_LOG")) {   // per wave of the last opaque VIS launch, as text
        static unsigned long long log[65536][4];
        if (hipMemcpyFromSymbol(log, HIP_SYMBOL(tr::tr_shade_wave_log), sizeof(log)) == hipSuccess) {
            if (FILE* fp = std::fopen(path, "w")) {
                for (int i = 0; i < 65536; ++i)
                    if (log[i][1]) std::fprintf(fp, "%d %llu %llu %llu %llu\n", i, log[i][0], log[i][1], log[i][2], log[i][3]);
                std::fclose(fp);
            }
        }
    }
    unsigned long long per_xcd[8] = {0};   // busy ticks per XCD (block b runs on XCD b % 8)
    for (int i = 0; i < 1024; ++i) per_xcd[i & 7] += host[3][i];
    out[6] = *std::max_element(per_xcd, per_xcd + 8) * 1000ull / (out[3] / 8ull + 1ull);   // most loaded XCD, per mille of the mean
    std::memset(host, 0, sizeof(host));
    return hipMemcpyToSymbol(HIP_SYMBOL(tr::tr_timing_counters), host, sizeof(host)) == hipSuccess ? 0 : -1;
}
#endif
#endif  // TR_PROBE_HOST
