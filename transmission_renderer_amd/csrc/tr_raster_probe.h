// tr_raster_probe.h — measurement scaffolding for raster_kernel.  NOT part of the product: tr_raster_kernels.h / tr_shade.hip
// include this file only under -DTR_RASTER_TIMING=1, which only tools/build_variant.py passes (build_ab/libtr_NAME.so;
// __graft_entry__.compile_library refuses such flags for transmission_renderer_amd/libtr_shade.so).
// Every wave of the opaque layer adds its phase times (s_memtime ticks) to tr_raster_timing[k][blockIdx & 1023] —
// 0 preparation, 1 item hand-over, 2 block loops, 3 total, 4 live items, 5 waves, 6 longest wave (max), 7 blocks visited,
// 8 issue time, 9 alpha-clipped blocks, 10 busiest workgroup's mean wave (max), 11 fragments — and writes a log line of its own
// (tr_raster_wave_log: total, preparation, block loops, packed counts, fragments, begin and end on the 100 MHz clock, hand-over).
// tools/gpu_bench_frame.py and tools/gpu_raster_stress.py print the sums; TR_WAVE_LOG=path / TR_TIMING_DUMP=1 dump the rest.
#ifndef TR_RASTER_PROBE_DEVICE
#define TR_RASTER_PROBE_DEVICE
// (included inside namespace tr)
__device__ unsigned long long tr_raster_timing[12][1024];
__device__ unsigned long long tr_raster_wave_log[8192][8];
#define TR_RT_NOW() __builtin_amdgcn_s_memtime()
#define TR_RT(x) x
#define TR_RT_WAVE_BEGIN \
    unsigned long long rt_search = 0, rt_pro = 0, rt_blocks = 0, rt_items = 0, rt_nblocks = 0, rt_frags = 0, rt_alpha = 0; \
    const unsigned long long rt_begin = TR_RT_NOW(), rt_real0 = __builtin_amdgcn_s_memrealtime();
#define TR_RT_WAVE_END \
    const unsigned long long rt_issue = TR_RT_NOW() - rt_begin; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    const unsigned long long rt_total = TR_RT_NOW() - rt_begin; \
    if (lane == 0u && layer == 0u && first < 8192u) { \
    unsigned long long* w = tr_raster_wave_log[first]; \
    w[0] = rt_total; w[1] = rt_search; w[2] = rt_blocks; w[3] = rt_items | (rt_nblocks << 16) | (rt_alpha << 32) | (rt_frags << 48 >> 48 << 48); \
    w[4] = rt_frags; w[5] = rt_real0; w[6] = __builtin_amdgcn_s_memrealtime(); w[7] = rt_pro; \
    } \
    if (lane == 0u && layer == 0u) { \
    const uint32_t s = blockIdx.x & 1023u; \
    atomicAdd(&tr_raster_timing[0][s], rt_search); atomicAdd(&tr_raster_timing[1][s], rt_pro); \
    atomicAdd(&tr_raster_timing[2][s], rt_blocks); atomicAdd(&tr_raster_timing[3][s], rt_total); \
    atomicAdd(&tr_raster_timing[4][s], rt_items);  atomicAdd(&tr_raster_timing[5][s], 1ull); \
    atomicMax(&tr_raster_timing[6][s], rt_total);  atomicAdd(&tr_raster_timing[7][s], rt_nblocks); \
    atomicAdd(&tr_raster_timing[8][s], rt_issue);  atomicAdd(&tr_raster_timing[9][s], rt_alpha); \
    atomicAdd(&tr_raster_timing[11][s], rt_frags); \
    } \
    __shared__ unsigned long long rt_wg[2];\
    if (threadIdx.x == 0u) { rt_wg[0] = 0ull; rt_wg[1] = 0ull; } \
    __syncthreads(); \
    if (lane == 0u) { atomicAdd(&rt_wg[0], rt_total - rt_search); atomicAdd(&rt_wg[1], rt_search); } \
    __syncthreads(); \
    if (threadIdx.x == 0u && layer == 0u) atomicMax(&tr_raster_timing[10][blockIdx.x & 1023u], rt_wg[0] / 4ull + rt_wg[1] / 4ull);
#endif  // TR_RASTER_PROBE_DEVICE

#ifdef TR_RASTER_PROBE_HOST
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
// (profiling builds only, not declared in include/) sums and clears the raster waves' phase counters of the opaque layer
extern "C" int32_t tr_debug_read_raster_timing(unsigned long long out[12]) {
    static unsigned long long host[12][1024];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(tr::tr_raster_timing), sizeof(host)) != hipSuccess) return -1;
    for (int k = 0; k < 12; ++k) {
        out[k] = 0;
        for (int i = 0; i < 1024; ++i) out[k] = (k == 6 || k == 10) ? std::max(out[k], host[k][i]) : out[k] + host[k][i];
    }
    if (const char* path = std::getenv("TR_WAVE_LOG")) {   // per wave: the raw log, as text
        static unsigned long long log[8192][8];
        if (hipMemcpyFromSymbol(log, HIP_SYMBOL(tr::tr_raster_wave_log), sizeof(log)) == hipSuccess) {
            if (FILE* fp = std::fopen(path, "w")) {
                for (int i = 0; i < 8192; ++i)
                    std::fprintf(fp, "%d %llu %llu %llu %llu %llu %llu %llu %llu\n", i, log[i][0], log[i][1], log[i][2], log[i][3], log[i][4], log[i][5], log[i][6], log[i][7]);
                std::fclose(fp);
            }
        }
    }
    if (std::getenv("TR_TIMING_DUMP")) {   // per block slot: mean wave ticks, blocks visited, live items
        for (int i = 0; i < 1024; ++i)
            std::fprintf(stderr, "slot %d %llu %llu %llu %llu %llu %llu %llu %llu\n", i, host[5][i] ? host[3][i] / host[5][i] : 0ull, host[7][i], host[4][i], host[5][i] ? host[0][i] / host[5][i] : 0ull, host[11][i], host[9][i], host[6][i], host[10][i]);
    }
    std::memset(host, 0, sizeof(host));
    return hipMemcpyToSymbol(HIP_SYMBOL(tr::tr_raster_timing), host, sizeof(host)) == hipSuccess ? 0 : -1;
}
#endif  // TR_RASTER_PROBE_HOST
