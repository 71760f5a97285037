// tr_shade.hip — C ABI (include/tr_shade.h) over the gfx950 kernels in tr_kernels.h.
//
// Host side of the boundary: owns the digested tables (materials, lights, GGX LUT pair table,
// pyramid level table) in HBM, validates arguments, and enqueues kernels on the caller's
// stream.  Nothing here computes pixels on the CPU; without a HIP device tr_context_create
// fails with TR_ERR_NO_DEVICE and every other entry point needs a context.
#if defined(TR_ABLATION) || defined(TR_TIMING) || defined(TR_PROBE_MASK)
#define TR_PROBE_HOST 1     // profiling builds (tools/build_variant.py): tr_probe.h's host hooks; empty in the product
#else
#define TR_PROBE_FRAME_PARAMS(fp)
#define TR_PROBE_GRID(bpx)
#endif
#include "tr_kernels.h"
#ifdef TR_TUNING_ENV   // tools/build_variant.py builds only: the loader / consumer variant of the plane pass (measured slower: docs/TRIED_r06.md)
#include "tr_lc_kernels.h"
#endif
#include "tr_cluster_kernels.h"
#include "tr_geometry_kernels.h"
#include "tr_raster_kernels.h"
#include "tr_glam_pbr_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: RCCL is loaded with dlopen (tr_comm_*), never linked

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace tr;

namespace tr {
// The frame recorder's first launch: frustum culling and light assignment are independent of each other, and each is
// a few microseconds of launch latency on its own.  Blocks [0, cull_blocks) cull, the next assign_blocks assign lights,
// the rest zero the rasteriser's tile coverage maps and list counters (a fill of their own is two more dispatches).
// The culling block that FINISHES LAST also demultiplexes the draws and scans both layers' draw streams (one workgroup's
// work that used to be the next launch): a ticket counts the culling blocks; what is handed over — the instance counts —
// was written by atomics and is read back by agent-scope loads, so no fence is involved (a device-scope release writes
// back the calling XCD's whole L2 on this chip; round 3's fence-and-ticket version of this launch took 115 us).
struct tr_front_demux {
    uint32_t* ticket;              // zero between frames
    const tr_primitive_info* primitives;
    uint32_t num_primitives;
    uint32_t* draw_counts;
    tr_draw_buffers out;
    tr_two_layers two;
};
__global__ __launch_bounds__(256) void frame_front_kernel(const tr_cull_params cp, const tr_primitive_info* __restrict__ primitives,
                                                          const tr_instance* __restrict__ instances,
                                                          uint32_t* __restrict__ instance_counts, uint32_t cull_blocks,
                                                          const tr_assign_params ap, const tr_alight* __restrict__ lights,
                                                          const tr_cluster_aabb* __restrict__ clusters,
                                                          uint32_t* __restrict__ cluster_counts, uint32_t* __restrict__ light_indices,
                                                          uint32_t assign_blocks, uint4* __restrict__ clear, uint32_t clear_vectors,
                                                          const tr_front_demux dm) {
    if (blockIdx.x < cull_blocks) {
        frustum_culling_body(cp, primitives, instances, instance_counts, blockIdx.x);
        __shared__ uint32_t last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wave's count atomics are performed ...
        __syncthreads();                                     //  ... and every other wave's of the block)
        if (threadIdx.x == 0u)
            last = __hip_atomic_fetch_add(dm.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cull_blocks - 1u ? 1u : 0u;
        __syncthreads();
        if (last == 0u) return;
        demultiplex_draws_body<true, 256u, true>(dm.primitives, instance_counts, dm.num_primitives, dm.draw_counts, dm.out);
        __threadfence_block();     // (the draw scans below read what this workgroup has just written)
        __syncthreads();
        const volatile uint32_t* counts_now = dm.draw_counts;
        for (uint32_t layer = 0; layer < 2u; ++layer) {
            TR_PICK_LAYER(dm.two, layer);
            if (W.capacity_triangles != 0u)
                raster_scan_draws_body<256u>(W.draws_a, W.draws_b, const_cast<const uint32_t*>(counts_now), W.buffer_a, dm.num_primitives,
                                             W.capacity_triangles, W.tri_base, W.counts);
            __syncthreads();
        }
        if (threadIdx.x == 0u) __hip_atomic_store(dm.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (blockIdx.x < cull_blocks + assign_blocks) {
        assign_lights_body(ap, lights, clusters, cluster_counts, light_indices, blockIdx.x - cull_blocks);
    } else {
        const uint32_t i = (blockIdx.x - cull_blocks - assign_blocks) * 256u + threadIdx.x;
        if (i < clear_vectors) clear[i] = uint4{0u, 0u, 0u, 0u};
    }
}
}  // namespace tr

namespace tr {
// Zero fill of `words` 32-bit words.  The per-frame clears go through this kernel, not hipMemsetAsync: a memset node of a
// captured frame wrote the words it should have cleared with pointer-like garbage from its second replay on (ROCm 7.0's HIP
// runtime under torch 2.10; tools/gpu_debug_capture2.py), a kernel node replays like any other launch.
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, uint32_t words) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) p[i] = 0u;
}
}  // namespace tr

struct tr_context {
    int device = 0;
    int32_t last_hip_error = 0;
    uint32_t blocks_per_xcd = 1024;  // (CUs / 8) * resident blocks per CU * kGridRounds, set at context creation
    uint32_t lc_wgs_per_cu = 0;      // > 0: the untextured transmissive plane pass runs as loader / consumer workgroups (tr_lc_kernels.h)

    // materials
    tr_material_info* d_materials_raw = nullptr;
    tr_dmat* d_dmats = nullptr;
    tr_dtap* d_dtaps = nullptr;     // per material: the refraction tap's level pair (digest_taps_kernel), valid for ...
    bool dtaps_valid = false;       // ... these digested materials, the level table in d_levels and this framebuffer width
    float dtaps_log2_width = 0.0f;
    uint32_t num_materials = 0, cap_materials = 0;
    bool dmats_dirty = false;
    std::vector<tr_material_info> stage_materials;

    bool any_textured = false;      // some material has a texture slot: the launches are the kTexLite build ...
    bool any_full_textured = false; // ... or, with a material outside the lite class, kTexAll / kTexAllMid (tr_kernels.h, TEX)
    uint32_t full_slots = 0;        // the texture slots bound by any full-class material (bit k: slot k of the kernels' `ids`)
    int32_t max_texture_id = -1;    // largest texture id a material refers to

    // material textures: one arena of RGBA8 mip chains + a descriptor table
    uint32_t* d_tex_arena = nullptr;
    tr_dtex* d_textures = nullptr;
    uint32_t num_textures = 0;
    std::vector<tr_dtex> h_textures;
    tr_colour_tables* d_colour_tables = nullptr;

    // geometry (model buffers) + the rasteriser's per-frame work buffers
    float* d_position = nullptr;
    float* d_normal = nullptr;
    float* d_uv = nullptr;
    uint32_t* d_index = nullptr;
    tr_primitive_info* d_primitives = nullptr;
    tr_instance* d_instances = nullptr;
    uint32_t num_vertices = 0, num_indices = 0, num_primitives = 0, num_instances = 0;
    std::vector<uint32_t> h_instance_primitive;   // primitive_id of every uploaded instance (tr_update_instances keeps it)
    uint32_t max_triangles[2] = {0, 0};       // per layer, if every instance is visible
    size_t work_capacity = 0, work_draws = 0; // elements per layer in the rasteriser's work buffers
    uint32_t* d_instance_counts = nullptr;
    uint32_t* d_front_ticket = nullptr;       // the frame's first launch: culling blocks done (zero between frames)
    uint32_t* d_draw_counts = nullptr;
    tr_draw_command* d_draws[TR_NUM_DRAW_BUFFERS] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t* d_tri_base = nullptr;
    tr_tri_record* d_records = nullptr;
    tr_tri_planes* d_tri_planes = nullptr;   // per triangle, beside the record: what the shading launches interpolate from
    uint32_t* d_item_base = nullptr;
    unsigned long long* d_scan_status = nullptr;   // per layer and set-up workgroup: the look-back words of the work-item prefix
    uint32_t scan_blocks = 0;                      // ... workgroups per layer (the words' stride)
    tr_layer_counts* d_layer_counts = nullptr;
    unsigned long long* d_vis[2] = {nullptr, nullptr};
    uint32_t* d_tile_cover[2] = {nullptr, nullptr};   // per layer: one word per 64x4 block tile (inside the d_vis allocation)
    uint32_t* d_front_list_count = nullptr;            // behind the maps: how many block tiles hold transmissive fragments, and which
    uint32_t* d_front_list = nullptr;                  //   (listed by the opaque VIS launch, walked by the transmissive one; kFrontLists sub-lists)
    uint32_t front_list_cap = 0;
    bool front_list_hint = false;                      // set by tr_record_frame around its VIS shading calls
    const uint32_t* cover_hint = nullptr;              // set by tr_record_frame around its shading calls only
    bool cover_cleared = false;                        // the frame's first launch has zeroed the coverage maps already
    unsigned long long* vis_hint = nullptr;            // ... and, when the frame skipped the resolve, the layer's visibility
    const tr_tri_planes* planes_hint = nullptr;        //     words and triangle planes: the shading launches read those (VIS)
    unsigned long long* vis_front_hint = nullptr;      // opaque VIS launches: the transmissive layer's words and coverage map
    const uint32_t* cover_front_hint = nullptr;        //   (a transmissive winner behind the opaque surface is zeroed there)
    uint32_t* present_hint = nullptr;                  // ... and the tonemapped frame its launches write final pixels to
    tr_tonemap_params present_params_hint = {};
    int32_t present_bgra_hint = 0;
    void* mip1_hint = nullptr;                         // ... and level 1 of the opaque pyramid, for the opaque launches to write
    size_t vis_pixels = 0;
    bool vis_clean = false;                            // both visibility buffers are all zero (stream order)
    uint32_t vis_w = 0, vis_h = 0;                     // the frame size they are laid out for
    bool counts_clean = false;                         // d_instance_counts is all zero (stream order): the fused frame path
    uint32_t num_cus = 256;
    uint32_t strip_rows = 0, strip_world = 1, strip_rank = 0;   // tr_set_strips: whole-frame shading calls take this rank's strips
    uint32_t tap_row_lo = 0, tap_row_hi = 0;                    // tr_set_tap_window: the pyramid rows (levels 0, 1) this rank holds
    uint32_t* tap_excess = nullptr;                             // ... and the device word the transmissive pass reports excursions in
    bool occupancy_fallback = false;     // the occupancy query failed: the grid was sized for 8 waves per SIMD
    bool mip_tail_attr_set = false;      // hipFuncSetAttribute(mip_tail_kernel, 160 KiB of LDS) done on this context's device

    // lights (as the shading kernels and as the cluster assignment read them)
    tr_dlight* d_lights = nullptr;
    tr_alight* d_alights = nullptr;
    uint32_t num_lights = 0, cap_lights = 0;
    std::vector<tr_dlight> stage_lights;
    std::vector<tr_alight> stage_alights;

    // cluster tables (borrowed)
    const uint32_t* d_cluster_counts = nullptr;
    const uint32_t* d_light_indices = nullptr;
    uint32_t num_clusters_total = 0;

    // GGX LUT
    uint32_t* d_lut_rgba8 = nullptr;
    uint32_t* d_lut_pairs = nullptr;
    float4* d_lut_lines = nullptr;      // per material, see build_lut_lines_kernel
    size_t lut_lines_entries = 0;
    uint32_t lut_w = 0, lut_h = 0, lut_stride = 0;
    std::vector<uint8_t> stage_lut;

    // pyramid level table
    tr_level_table* d_levels = nullptr;
    tr_level_table h_levels{};
    uint32_t h_levels_count = 0;

    // cluster x / y lookup tables (exact u32(frag_coord / cluster_size), shader/src/lib.rs:89)
    uint32_t* d_cluster_x = nullptr;
    uint32_t* d_cluster_y_term = nullptr;
    uint32_t cl_w = 0, cl_h = 0, cl_cap_w = 0, cl_cap_h = 0, cl_ncx = 0;
    float cl_sx = 0.0f, cl_sy = 0.0f;
    std::vector<uint32_t> stage_cluster_x;
    std::vector<uint32_t> stage_cluster_y;

    // depth-slice thresholds of the bound LightClusterCoefficients (build_slice_thresholds)
    float* d_slice_thr = nullptr;
    tr_light_cluster_coefficients slice_coeffs{};
    bool slice_valid = false;
    uint32_t slice_max = 0;
    float h_slice_thr[TR_MAX_DEPTH_SLICES + 2]{};

    // The lazily built device tables (digested materials, LUT lines, level table, tap records, cluster x / y tables, slice
    // thresholds) are enqueued on the stream of the call that finds them stale.  A context may be driven from several
    // streams (a pass as two bands on two streams): a call on ANOTHER stream waits for `tables_event`
    // (recorded behind the last build) before it launches, and a rebuild first waits for every stream that launched
    // against the previous tables (tables_before_rebuild).
    hipEvent_t tables_event = nullptr;
    hipStream_t tables_stream = nullptr;
    uint64_t tables_generation = 0;
    struct stream_seen { hipStream_t stream; uint64_t generation; };
    std::vector<stream_seen> launch_streams;   // streams that launched since the last build, and the build they waited for
    uint32_t vis_grid_rounds = 6;               // see persistent_grid (TR_VIS_ROUNDS in tools/ builds)
    uint32_t front_list_waves_per_cu = 48;      // the transmissive VIS launch's grid when it walks the list of covered tiles (TR_FRONT_LIST_WAVES in tools/ builds; 0: no list)
    uint32_t raster_wgs_per_cu = 6;             // raster_kernel's persistent grid: 4 -> 191 / 199 us (4K mesh / glTF demo frame), 6 -> 191 / 192,
                                                // 8 -> 194 / 195, 12 -> 191 / 196, 16 -> 198 / 199 (TR_RASTER_WGS_PER_CU in tools/ builds)
};

namespace {

inline hipError_t zero_fill(void* p, size_t bytes, hipStream_t stream) {   // (bytes: a multiple of 4, below 16 GiB)
    const uint32_t words = (uint32_t)(bytes / 4u);
    if (words == 0u) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<size_t>((words + 255u) / 256u, 4096u);
    hipLaunchKernelGGL(tr::zero_words_kernel, dim3(blocks), dim3(256), 0, stream, (uint32_t*)p, words);
    return hipGetLastError();
}

#define TR_HIP(ctx, expr)                                   \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                             \
            if (ctx) (ctx)->last_hip_error = (int32_t)e_;   \
            return TR_ERR_HIP;                              \
        }                                                   \
    } while (0)

inline uint32_t level_dim(uint32_t d, uint32_t l) { return (d >> l) ? (d >> l) : 1u; }

// mip_levels_for_size, src/main.rs:2590-2592: (min(w,h) as f32).log2() as u32 + 1
uint32_t mip_levels_for_size(uint32_t w, uint32_t h) {
    uint32_t m = w < h ? w : h;
    float l = std::log2((float)m);
    uint32_t li = (l > 0.0f) ? (uint32_t)l : 0u;
    return li + 1u;
}

// Before a lazily built table is overwritten on `stream`: launches enqueued on other streams may still read it.
// (Rebuilds are rare — new materials, another pyramid geometry, other cluster coefficients — so this is a device-wide
// wait, not an event per launch.)
tr_status tables_before_rebuild(tr_context* ctx, hipStream_t stream) {
    // (a table rebuild allocates, copies from pageable host memory or synchronises: none of it can be captured.  A frame is
    //  captured after one has run outside the capture, when every table is in place)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) return TR_ERR_UNSUPPORTED;
    bool others = false;
    for (const auto& s : ctx->launch_streams) others |= s.stream != stream;
    if (others) TR_HIP(ctx, hipDeviceSynchronize());
    ctx->launch_streams.clear();
    return TR_OK;
}
// After table work was enqueued on `stream`.
tr_status tables_rebuilt(tr_context* ctx, hipStream_t stream) {
    if (!ctx->tables_event) TR_HIP(ctx, hipEventCreateWithFlags(&ctx->tables_event, hipEventDisableTiming));
    TR_HIP(ctx, hipEventRecord(ctx->tables_event, stream));
    ctx->tables_stream = stream;
    ctx->tables_generation += 1u;
    return TR_OK;
}
// Before a launch on `stream` that reads the tables: ordered behind the build when that ran on another stream.
tr_status tables_acquire(tr_context* ctx, hipStream_t stream) {
    for (auto& s : ctx->launch_streams) {
        if (s.stream != stream) continue;
        if (s.generation == ctx->tables_generation) return TR_OK;
        if (stream != ctx->tables_stream && ctx->tables_event) TR_HIP(ctx, hipStreamWaitEvent(stream, ctx->tables_event, 0));
        s.generation = ctx->tables_generation;
        return TR_OK;
    }
    if (stream != ctx->tables_stream && ctx->tables_event) TR_HIP(ctx, hipStreamWaitEvent(stream, ctx->tables_event, 0));
    ctx->launch_streams.push_back({stream, ctx->tables_generation});
    return TR_OK;
}
#define TR_TRY(expr)                         \
    do {                                     \
        const tr_status st_ = (expr);        \
        if (st_ != TR_OK) return st_;        \
    } while (0)

tr_status ensure_digested(tr_context* ctx, hipStream_t stream) {
    if (!ctx->dmats_dirty) return TR_OK;
    if (ctx->num_materials == 0 || ctx->lut_h == 0) return TR_ERR_TABLES_MISSING;
    TR_TRY(tables_before_rebuild(ctx, stream));
    uint32_t n = ctx->num_materials;
    hipLaunchKernelGGL(digest_materials_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, ctx->d_materials_raw,
                       ctx->d_dmats, n, ctx->lut_h, ctx->lut_stride);
    const size_t entries = (size_t)n * ctx->lut_stride;
    if (entries * 16u > 0xFFFFFFFFull) return TR_ERR_UNSUPPORTED;
    if (entries > ctx->lut_lines_entries) {
        TR_HIP(ctx, hipStreamSynchronize(stream));
        (void)hipFree(ctx->d_lut_lines);
        ctx->d_lut_lines = nullptr;
        ctx->lut_lines_entries = 0;
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_lut_lines, entries * sizeof(float4)));
        ctx->lut_lines_entries = entries;
    }
    hipLaunchKernelGGL(build_lut_lines_kernel, dim3((ctx->lut_stride + 255u) / 256u, n), dim3(256), 0, stream,
                       (const uint32_t*)ctx->d_lut_pairs, (const tr_dmat*)ctx->d_dmats, ctx->d_lut_lines, ctx->lut_stride, n);
    TR_HIP(ctx, hipGetLastError());
    ctx->dmats_dirty = false;
    ctx->dtaps_valid = false;
    return tables_rebuilt(ctx, stream);
}

tr_status ensure_levels(tr_context* ctx, const tr_pyramid* p, hipStream_t stream) {
    bool same = ctx->h_levels_count == p->levels;
    for (uint32_t l = 0; same && l < p->levels; ++l)
        same = ctx->h_levels.offset[l] == p->level_offset[l] && ctx->h_levels.width[l] == level_dim(p->width, l) &&
               ctx->h_levels.height[l] == level_dim(p->height, l);
    if (same) return TR_OK;
    TR_TRY(tables_before_rebuild(ctx, stream));
    std::memset(&ctx->h_levels, 0, sizeof(ctx->h_levels));
    for (uint32_t l = 0; l < p->levels; ++l) {
        ctx->h_levels.offset[l] = p->level_offset[l];
        ctx->h_levels.width[l] = level_dim(p->width, l);
        ctx->h_levels.height[l] = level_dim(p->height, l);
        ctx->h_levels.wf[l] = (float)ctx->h_levels.width[l];
        ctx->h_levels.hf[l] = (float)ctx->h_levels.height[l];
        ctx->h_levels.xlim[l] = ctx->h_levels.width[l] >= 2u ? (float)(ctx->h_levels.width[l] - 2u) : 0.0f;
    }
    ctx->h_levels_count = p->levels;
    ctx->dtaps_valid = false;
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_levels, &ctx->h_levels, sizeof(tr_level_table), hipMemcpyHostToDevice, stream));
    return tables_rebuilt(ctx, stream);
}

// The per-material tap records of the transmissive pass (tr_dtap): after ensure_digested and ensure_levels.
tr_status ensure_tap_records(tr_context* ctx, float log2_fb_width, hipStream_t stream) {
    if (ctx->dtaps_valid && std::memcmp(&ctx->dtaps_log2_width, &log2_fb_width, sizeof(float)) == 0) return TR_OK;
    if (!ctx->d_dtaps || ctx->num_materials == 0 || ctx->h_levels_count == 0) return TR_ERR_TABLES_MISSING;
    if ((uint64_t)ctx->h_levels.offset[ctx->h_levels_count - 1u] * 8u + 16u > 0xFFFFFFFFull) return TR_ERR_UNSUPPORTED;
    TR_TRY(tables_before_rebuild(ctx, stream));
    hipLaunchKernelGGL(digest_taps_kernel, dim3((ctx->num_materials + 63u) / 64u), dim3(64), 0, stream,
                       (const tr_dmat*)ctx->d_dmats, (const tr_level_table*)ctx->d_levels, ctx->h_levels_count, log2_fb_width,
                       ctx->d_dtaps, ctx->num_materials);
    TR_HIP(ctx, hipGetLastError());
    ctx->dtaps_valid = true;
    ctx->dtaps_log2_width = log2_fb_width;
    return tables_rebuilt(ctx, stream);
}

// The sRGB transfer functions of the Khronos data format specification (13.3), with the host's libm.
inline float srgb_to_linear(uint32_t byte) {
    const float x = (float)byte / 255.0f;
    return x <= 0.04045f ? x / 12.92f : std::pow((x + 0.055f) / 1.055f, 2.4f);
}
inline uint32_t linear_to_srgb8(float x) {
    if (!(x > 0.0f)) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    const float e = x <= 0.0031308f ? 12.92f * x : 1.055f * std::pow(x, 1.0f / 2.4f) - 0.055f;
    return (uint32_t)(e * 255.0f + 0.5f);
}

// The decode tables and, for the encode, the smallest float that maps to each byte (found by bisection over the
// bit patterns of [0, 1]: the encode is monotonic), so the device reproduces the host function with compares.
void fill_colour_tables(tr_colour_tables* t) {
    for (uint32_t b = 0; b < 256u; ++b) {
        t->srgb_to_linear[b] = srgb_to_linear(b);
        t->unorm[b] = (float)b / 255.0f;
    }
    t->srgb_threshold[0] = 0.0f;
    for (uint32_t b = 1; b < 256u; ++b) {
        uint32_t lo = 0u, hi = 0x3F800000u;   // bits of 0.0f .. 1.0f; invariant: enc(lo) < b <= enc(hi)
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            float f;
            std::memcpy(&f, &mid, 4);
            if (linear_to_srgb8(f) >= b) hi = mid;
            else lo = mid;
        }
        std::memcpy(&t->srgb_threshold[b], &hi, 4);
    }
}

// Rust `f32 as u32`: saturating, NaN -> 0
inline uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// cluster_xy = (frag_coord.xy / cluster_size_in_pixels).as_uvec2() (shader/src/lib.rs:89) for every
// column / row of the frame, with the reference's own IEEE fp32 division, done once per geometry.
tr_status ensure_cluster_tables(tr_context* ctx, const tr_uniforms* u, uint32_t fw, uint32_t fh, hipStream_t stream) {
    const float sx = u->cluster_size_in_pixels[0], sy = u->cluster_size_in_pixels[1];
    const uint32_t ncx = u->num_clusters[0];
    if (ctx->d_cluster_x && ctx->cl_w == fw && ctx->cl_h == fh && ctx->cl_sx == sx && ctx->cl_sy == sy &&
        ctx->cl_ncx == ncx)
        return TR_OK;
    TR_TRY(tables_before_rebuild(ctx, stream));
    if (fw > ctx->cl_cap_w) {
        (void)hipFree(ctx->d_cluster_x);
        ctx->d_cluster_x = nullptr;
        ctx->cl_cap_w = 0;
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_cluster_x, sizeof(uint32_t) * fw));
        ctx->cl_cap_w = fw;
    }
    if (fh > ctx->cl_cap_h) {
        (void)hipFree(ctx->d_cluster_y_term);
        ctx->d_cluster_y_term = nullptr;
        ctx->cl_cap_h = 0;
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_cluster_y_term, sizeof(uint32_t) * fh));
        ctx->cl_cap_h = fh;
    }
    ctx->stage_cluster_x.resize(fw);
    ctx->stage_cluster_y.resize(fh);
    for (uint32_t x = 0; x < fw; ++x) {
        uint32_t c = f32_as_u32(((float)x + 0.5f) / sx);
        ctx->stage_cluster_x[x] = c > 0xFFFFu ? 0xFFFFu : c;
    }
    for (uint32_t y = 0; y < fh; ++y) ctx->stage_cluster_y[y] = f32_as_u32(((float)y + 0.5f) / sy) * ncx;
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_cluster_x, ctx->stage_cluster_x.data(), sizeof(uint32_t) * fw,
                               hipMemcpyHostToDevice, stream));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_cluster_y_term, ctx->stage_cluster_y.data(), sizeof(uint32_t) * fh,
                               hipMemcpyHostToDevice, stream));
    ctx->cl_w = fw;
    ctx->cl_h = fh;
    ctx->cl_sx = sx;
    ctx->cl_sy = sy;
    ctx->cl_ncx = ncx;
    return tables_rebuilt(ctx, stream);
}

// LightClusterCoefficients::get_depth_slice (shared-structs/src/lib.rs:43-63) with the reference's own fp32
// operations and the host's libm (what the oracle restates).
inline uint32_t depth_slice_reference(const tr_light_cluster_coefficients& c, float frag_depth) {
#pragma clang fp contract(off)
    const float depth_range = 2.0f * (1.0f - frag_depth) - 1.0f;
    const float linear = 2.0f * c.z_near * c.z_far / (c.z_far + c.z_near - depth_range * (c.z_far - c.z_near));
    return f32_as_u32(std::fmax(std::log2(linear) * c.scale + c.bias, 0.0f));
}

// thr[k] = the largest depth in [0, +inf] whose slice is >= k, by bisection over the bit patterns (the slice is a
// non-increasing step function of a non-negative depth); thr[0] = +inf, thr[slice_max + 1] = -1 (no depth >= 0 is
// below it).  slice(d) = #{k >= 1 : d <= thr[k]}: the kernels' depth_slice() corrects its estimate with two of them.
tr_status build_slice_thresholds(const tr_light_cluster_coefficients& c, float* thr, uint32_t* slice_max) {
    const uint32_t top = depth_slice_reference(c, 0.0f);
    if (top > TR_MAX_DEPTH_SLICES) return TR_ERR_UNSUPPORTED;
    const uint32_t inf_bits = 0x7F800000u;
    std::memcpy(&thr[0], &inf_bits, 4);
    for (uint32_t k = 1; k <= top; ++k) {
        uint32_t lo = 0u, hi = inf_bits;   // slice(lo) >= k > slice(hi)   (slice(+inf) = 0)
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            float d;
            std::memcpy(&d, &mid, 4);
            if (depth_slice_reference(c, d) >= k) lo = mid;
            else hi = mid;
        }
        std::memcpy(&thr[k], &lo, 4);
    }
    thr[top + 1u] = -1.0f;
    *slice_max = top;
    return TR_OK;
}

tr_status ensure_slice_thresholds(tr_context* ctx, const tr_uniforms* u, hipStream_t stream) {
    const tr_light_cluster_coefficients& c = u->light_clustering_coefficients;
    if (ctx->slice_valid && std::memcmp(&ctx->slice_coeffs, &c, 20) == 0) return TR_OK;   // (the five fields, not the padding)
    uint32_t top = 0;
    const tr_status st = build_slice_thresholds(c, ctx->h_slice_thr, &top);
    if (st != TR_OK) return st;
    // (stream order protects this stream's launches that still read the previous table; other streams': below)
    TR_TRY(tables_before_rebuild(ctx, stream));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_slice_thr, ctx->h_slice_thr, sizeof(float) * (top + 2u), hipMemcpyHostToDevice, stream));
    ctx->slice_coeffs = c;
    ctx->slice_max = top;
    ctx->slice_valid = true;
    return tables_rebuilt(ctx, stream);
}

tr_status fill_frame_params(const tr_context* ctx, const tr_gbuffer* g, const tr_uniforms* u,
                            const tr_push_constants* pc, tr_rect rect, tr_frame_params* fp) {
    const uint32_t fw = pc->framebuffer_size[0], fh = pc->framebuffer_size[1];
    if (rect.x0 >= rect.x1 || rect.y0 >= rect.y1 || rect.x1 > fw || rect.y1 > fh) return TR_ERR_INVALID_ARGUMENT;
    if (rect.x0 < g->origin_x || rect.y0 < g->origin_y || rect.x1 > g->origin_x + g->width ||
        rect.y1 > g->origin_y + g->height)
        return TR_ERR_INVALID_ARGUMENT;  // the tile must be covered by the planes this rank holds
    // the kernels address every buffer as base + 32-bit byte offset and multiply coordinates with 24-bit operands
    if ((uint64_t)fw * fh * 16u > 0xFFFFFFFFull || (uint64_t)g->width * g->height * 16u > 0xFFFFFFFFull ||
        fw >= (1u << 24) || fh >= (1u << 24) || (uint64_t)ctx->num_clusters_total * TR_MAX_LIGHTS_PER_CLUSTER * 4u > 0xFFFFFFFFull)
        return TR_ERR_UNSUPPORTED;
    std::memset(fp, 0, sizeof(*fp));
    std::memcpy(fp->proj_view, pc->proj_view, sizeof(fp->proj_view));
    std::memcpy(fp->view_position, pc->view_position, sizeof(fp->view_position));
    fp->log2_fb_width = std::log2((float)pc->framebuffer_size[0]);  // glam-pbr/src/lib.rs:334
    std::memcpy(fp->sun_dir, u->sun_dir, sizeof(fp->sun_dir));
    std::memcpy(fp->sun_intensity, u->sun_intensity, sizeof(fp->sun_intensity));
    {
        // get_depth_slice (shared-structs/src/lib.rs:54-63): log2(linear) * scale + bias = K - scale * log2(denominator);
        // the kernels form the denominator with the reference's own roundings and settle the integer against
        // the threshold table (depth_slice in tr_kernels.h)
        const tr_light_cluster_coefficients& c = u->light_clustering_coefficients;
        const double n = c.z_near, f = c.z_far;
        fp->lcc_scale = c.scale;
        fp->slice_fpn = c.z_far + c.z_near;
        fp->slice_fmn = c.z_far - c.z_near;
        fp->slice_k = (float)(std::log2(2.0 * n * f) * (double)c.scale + (double)c.bias);
        fp->slice_max = ctx->slice_max;
    }
    fp->clusters_xy = u->num_clusters[0] * u->num_clusters[1];
    fp->num_clusters_total = ctx->num_clusters_total;
    fp->debug_clusters = u->debug_clusters;
    fp->width = fw;
    fp->height = fh;
    fp->g_width = g->width;
    fp->g_origin_x = g->origin_x;
    fp->g_origin_y = g->origin_y;
    fp->rect_x0 = rect.x0;
    fp->rect_y0 = rect.y0;
    fp->rect_x1 = rect.x1;
    fp->rect_y1 = rect.y1;
    fp->tiles_x = (rect.x1 - rect.x0 + kBlockTileW - 1u) / kBlockTileW;
    fp->tiles_y = (rect.y1 - rect.y0 + kBlockTileH - 1u) / kBlockTileH;
    if (ctx->strip_rows != 0u && ctx->strip_world > 1u) {
        // rank-interleaved strips: the rect must be the whole frame height; the launch sweeps this rank's tile rows only
        if (rect.y0 != 0u || rect.y1 != fh) return TR_ERR_INVALID_ARGUMENT;
        const uint32_t T = ctx->strip_rows / kBlockTileH;
        const uint32_t frame_tile_rows = fp->tiles_y, strips = (frame_tile_rows + T - 1u) / T;
        const uint32_t owned = ctx->strip_rank < strips ? (strips - 1u - ctx->strip_rank) / ctx->strip_world + 1u : 0u;
        uint32_t rows = owned * T;
        if (owned != 0u && (strips - 1u) % ctx->strip_world == ctx->strip_rank) rows -= strips * T - frame_tile_rows;   // the frame's last strip is short
        fp->tiles_y = rows;
        fp->strip_tile_rows = T;
        fp->strip_magic = T == 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / T);   // (2^32 / 1 does not fit: clamped like the other magics)
        fp->strip_world = ctx->strip_world;
        fp->strip_rank = ctx->strip_rank;
    }
    fp->tap_row_lo = ctx->tap_row_lo;
    fp->tap_row_hi = ctx->tap_excess ? ctx->tap_row_hi : 0u;
    fp->tiles_x_magic = (uint32_t)(0x100000000ull / fp->tiles_x > 0xFFFFFFFFull ? 0xFFFFFFFFull : 0x100000000ull / fp->tiles_x);
    fp->stripe_tiles = fp->tiles_x * kStripeTileRows;
    fp->stripe_magic = (uint32_t)(0x100000000ull / fp->stripe_tiles > 0xFFFFFFFFull ? 0xFFFFFFFFull : 0x100000000ull / fp->stripe_tiles);
    TR_PROBE_FRAME_PARAMS(fp)
    fp->lut_wf = (float)ctx->lut_w;
    fp->lut_stride = ctx->lut_stride;
    fp->lut_height = ctx->lut_h;
    return TR_OK;
}

// Persistent grid: 8 XCDs x k blocks, k chosen so that every CU holds its 4 resident 256-thread blocks.
// `vis`: a launch of the frame recorder (visibility words, XCD stripes, most tiles of a real frame empty or cheap): fewer,
// longer-lived waves — kVisGridRounds instead of kGridRounds times what is resident.  4K mesh frame by rounds
// (TR_BLOCKS_PER_XCD sweep, all shading launches of the frame): 1 -> 243.8 us, 2 -> 227.0, 4 -> 227.0, 7 -> 228.9,
// 8 -> 232.1, 16 -> 247.0: a wave's start (kernarg loads, the sRGB table into LDS) is paid per wave, and a frame's covered
// tiles are spread evenly over the stripes anyway.  (End of round 4, frame at 172 us: 2 and 3 -> 172.2, 4 -> 170.6, 5 -> 172.2;
// glTF demo 176.0 / 173.9 / 173.6: 4.  Round 5, background tiles written from constants, so that a wave's tiles differ tenfold
// in cost: 2 / 3 / 4 / 5 / 6 / 7 / 8 / 12 / 16 -> 168.4 / 168.1 / 166.2 / 168.2 / 165.3 / 167.2 / 166.5 / 170.1 / 171.3 us, glTF demo 168.8 at 4,
// 165.6 at 6, 8K 641 / 632: 6.)  The synthetic G-buffer's plane launches keep their own count (kGridRounds, §3.1 of DESIGN.md).
uint32_t persistent_grid(const tr_context* ctx, uint32_t ntiles, bool vis = false) {
    const uint32_t per_xcd = (ntiles + 7u) / 8u;
    uint32_t bpx = vis ? ctx->blocks_per_xcd / kGridRounds * ctx->vis_grid_rounds : ctx->blocks_per_xcd;
    TR_PROBE_GRID(bpx)
    uint32_t k = bpx < per_xcd ? bpx : per_xcd;
    if (k == 0) k = 1;
    return 8u * k;
}

void free_geometry(tr_context* ctx) {
    (void)hipFree(ctx->d_position);
    (void)hipFree(ctx->d_normal);
    (void)hipFree(ctx->d_uv);
    (void)hipFree(ctx->d_index);
    (void)hipFree(ctx->d_primitives);
    (void)hipFree(ctx->d_instances);
    (void)hipFree(ctx->d_instance_counts);
    (void)hipFree(ctx->d_draw_counts);
    for (auto& d : ctx->d_draws) {
        (void)hipFree(d);
        d = nullptr;
    }
    (void)hipFree(ctx->d_tri_base);
    (void)hipFree(ctx->d_records);
    (void)hipFree(ctx->d_tri_planes);
    (void)hipFree(ctx->d_item_base);
    (void)hipFree(ctx->d_scan_status);
    (void)hipFree(ctx->d_layer_counts);
    ctx->d_position = ctx->d_normal = ctx->d_uv = nullptr;
    ctx->d_index = nullptr;
    ctx->d_primitives = nullptr;
    ctx->d_instances = nullptr;
    ctx->d_instance_counts = ctx->d_draw_counts = ctx->d_tri_base = ctx->d_item_base = nullptr;
    ctx->d_scan_status = nullptr;
    ctx->d_records = nullptr;
    ctx->d_tri_planes = nullptr;
    ctx->d_layer_counts = nullptr;
    ctx->num_vertices = ctx->num_indices = ctx->num_primitives = ctx->num_instances = 0;
    ctx->max_triangles[0] = ctx->max_triangles[1] = 0;
}

// Passes that shade textured materials differentiate inside 2x2 pixel quads: the rect must hold whole quads.
tr_status check_textured_launch(const tr_context* ctx, const tr_gbuffer* g, const tr_frame_params& fp) {
    if (!ctx->any_textured) return TR_OK;
    if (!g->uv) return TR_ERR_INVALID_ARGUMENT;
    if (ctx->max_texture_id >= (int32_t)ctx->num_textures) return TR_ERR_INVALID_ARGUMENT;
    if ((fp.rect_x0 & 1u) || (fp.rect_y0 & 1u)) return TR_ERR_INVALID_ARGUMENT;
    if (((fp.rect_x1 & 1u) && fp.rect_x1 != fp.width) || ((fp.rect_y1 & 1u) && fp.rect_y1 != fp.height))
        return TR_ERR_INVALID_ARGUMENT;
    return TR_OK;
}

void fill_launch(tr_launch& L, const tr_context* ctx, const tr_frame_params& fp, const tr_gbuffer* g) {
    L.fp = fp;
    L.dmats = ctx->d_dmats;
    L.lights = ctx->d_lights;
    L.cluster_counts = ctx->d_cluster_counts;
    L.light_indices = ctx->d_light_indices;
    L.lut_pairs = ctx->d_lut_pairs;
    L.lut_lines = ctx->d_lut_lines;
    L.levels = ctx->d_levels;
    L.dtaps = ctx->d_dtaps;
    // the coverage map of the layer, when tr_record_frame rasterised it itself and shades the whole frame
    L.tile_cover = (ctx->cover_hint && fp.rect_x0 == 0u && fp.rect_y0 == 0u && fp.g_origin_x == 0u && fp.g_origin_y == 0u &&
                    fp.rect_x1 == fp.width && fp.rect_y1 == fp.height && fp.g_width == fp.width)
                       ? ctx->cover_hint : nullptr;
    L.vis = L.tile_cover ? ctx->vis_hint : nullptr;
    L.tri_planes = L.vis ? ctx->planes_hint : nullptr;
    L.vis_front = L.vis ? ctx->vis_front_hint : nullptr;
    L.cover_front = L.vis ? ctx->cover_front_hint : nullptr;
    L.present = L.vis ? ctx->present_hint : nullptr;
    L.front_list_build = L.front_list_build_count = nullptr;
    L.front_list = L.front_list_count = nullptr;
    L.front_list_cap = 0u;
    L.present_params = ctx->present_params_hint;
    L.present_e1 = ctx->present_params_hint.saturation / ctx->present_params_hint.cross_saturation;   // (as tr_tonemap forms it)
    L.present_bgra = ctx->present_bgra_hint;
    L.tap_excess = ctx->tap_excess;
    L.slice_thr = ctx->d_slice_thr;
    L.cluster_x = ctx->d_cluster_x;
    L.cluster_y_term = ctx->d_cluster_y_term;
    L.pos_depth = (const float4*)g->pos_depth;
    L.nrm_scale = (const float4*)g->nrm_scale;
    L.material_id = (const uint32_t*)g->material_id;
    L.pyramid = nullptr;
    L.hdr = nullptr;
    L.mip0 = nullptr;
    L.mip1 = nullptr;
    L.uv = (const float2*)g->uv;
    L.materials = ctx->d_materials_raw;
    L.textures = ctx->d_textures;
    L.tex_arena = ctx->d_tex_arena;
    L.srgb_to_linear = ctx->d_colour_tables ? ctx->d_colour_tables->srgb_to_linear : nullptr;
}

// One launch of shade_kernel<TRANSMISSIVE, ., TEX, .>: RGBA16F or RGBA32F target; planes, or (RGBA16F inside the frame
// recorder) the rasteriser's visibility words.
template <bool TRANSMISSIVE, int TEX>
void launch_shade(const tr_launch& L_, bool half, dim3 grid, dim3 block, hipStream_t stream) {
    tr_launch L = L_;
    L.fp.j_step = L.front_list ? grid.x / kFrontLists : grid.x >> 3;
    if (half && L.vis) hipLaunchKernelGGL((shade_kernel<TRANSMISSIVE, uint2, TEX, true>), grid, block, 0, stream, L);
    else if (half) hipLaunchKernelGGL((shade_kernel<TRANSMISSIVE, uint2, TEX, false>), grid, block, 0, stream, L);
    else hipLaunchKernelGGL((shade_kernel<TRANSMISSIVE, float4, TEX, false>), grid, block, 0, stream, L);
}

template <bool TRANSMISSIVE>
void launch_textured(tr_context* ctx, const tr_launch& L, bool half, dim3 grid, dim3 block, hipStream_t stream) {
    // ONE launch whatever classes the uploaded materials mix (tr_kernels.h, TEX).  With a full-class material: its kTexAllMid
    // build when none binds a slot beyond base colour, metallic-roughness and normal map (the transmission / thickness
    // slots do not exist for the opaque pass)
    if (!ctx->any_full_textured) return launch_shade<TRANSMISSIVE, kTexLite>(L, half, grid, block, stream);
    const bool mid = (ctx->full_slots & ~(TRANSMISSIVE ? kSlotsMid : (kSlotsMid | 0x30u))) == 0u;
    if (mid) launch_shade<TRANSMISSIVE, kTexAllMid>(L, half, grid, block, stream);
    else launch_shade<TRANSMISSIVE, kTexAll>(L, half, grid, block, stream);
}

bool tables_ready(const tr_context* ctx, bool need_lut) {
    return ctx->num_materials > 0 && ctx->d_cluster_counts && ctx->d_light_indices && ctx->num_clusters_total > 0 &&
           (!need_lut || ctx->d_lut_pairs) && ctx->lut_h > 0;
}

}  // namespace

extern "C" {

uint32_t tr_abi_version(void) { return TR_ABI_VERSION; }

const char* tr_status_string(tr_status status) {
    switch (status) {
        case TR_OK: return "ok";
        case TR_ERR_INVALID_ARGUMENT: return "invalid argument";
        case TR_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
        case TR_ERR_HIP: return "HIP runtime error (see tr_last_hip_error)";
        case TR_ERR_TABLES_MISSING: return "pass launched before its tables were uploaded";
        case TR_ERR_OUT_OF_MEMORY: return "out of memory";
        case TR_ERR_UNSUPPORTED: return "unsupported";
        case TR_ERR_COMM: return "RCCL unavailable or a collective failed (see tr_comm_last_error)";
        default: return "unknown status";
    }
}

int32_t tr_last_hip_error(const tr_context* ctx) { return ctx ? ctx->last_hip_error : 0; }

tr_status tr_context_create(int32_t device_ordinal, tr_context** out_ctx) {
    if (!out_ctx) return TR_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TR_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= count) return TR_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device_ordinal) != hipSuccess) return TR_ERR_NO_DEVICE;
    tr_context* ctx = new (std::nothrow) tr_context();
    if (!ctx) return TR_ERR_OUT_OF_MEMORY;
    ctx->device = device_ordinal;
    {
        // Grid of the shading kernels: kGridRounds times the waves one CU holds (measured on MI355X, 8 waves per SIMD:
        // 1 / 2 / 4 / 6 / 8 / 32 rounds -> 117 / 115 / 112 / 111 / 111 / 117 us in the profiling build: the uneven ends
        // of the waves' runs are spread over more, shorter runs), 1/8 of them per XCD.  The occupancy is asked for the
        // block size that is launched (one-wave workgroups: 64 threads -> resident WAVES per CU; 32 when the kernel
        // holds 8 waves per SIMD); blocks_per_xcd counts units of four waves (persistent_grid's callers multiply by 4).
        // The TEXTURED variants hold fewer waves (4 per SIMD) and are launched with the same grid: 16 rounds for them.
        hipDeviceProp_t prop;
        int resident_waves = 0;
        const int launched_block = 64;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident_waves, shade_kernel<true, uint2>, launched_block, 0) != hipSuccess ||
            resident_waves <= 0) {
            resident_waves = 32 * 64 / launched_block;   // 8 waves per SIMD (the kernel is built for 64 VGPRs)
            ctx->occupancy_fallback = true;
        }
        const int resident = resident_waves * launched_block / 256 > 0 ? resident_waves * launched_block / 256 : 1;
        if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount >= 8) {
            ctx->blocks_per_xcd = (uint32_t)(prop.multiProcessorCount / 8) * (uint32_t)resident * kGridRounds;
            ctx->num_cus = (uint32_t)prop.multiProcessorCount;
        }
#ifdef TR_TUNING_ENV   // tools/build_variant.py builds only (build_ab/): the product reads nothing from its caller's environment
        if (const char* e = std::getenv("TR_BLOCKS_PER_XCD")) ctx->blocks_per_xcd = (uint32_t)std::atoi(e);
#ifndef TR_LC_DISABLE   // (an A/B's base build ignores the switch its siblings read)
        if (const char* e = std::getenv("TR_LC")) ctx->lc_wgs_per_cu = (uint32_t)std::max(0, std::atoi(e));
#endif
        if (const char* e = std::getenv("TR_VIS_ROUNDS")) ctx->vis_grid_rounds = (uint32_t)std::max(1, std::atoi(e));
        if (const char* e = std::getenv("TR_FRONT_LIST_WAVES")) ctx->front_list_waves_per_cu = (uint32_t)std::max(0, std::atoi(e));
        if (const char* e = std::getenv("TR_RASTER_WGS_PER_CU")) ctx->raster_wgs_per_cu = (uint32_t)std::max(1, std::atoi(e));
#endif
    }
    if (hipMalloc((void**)&ctx->d_front_ticket, 4u) != hipSuccess || hipMemset(ctx->d_front_ticket, 0, 4u) != hipSuccess ||
        hipMalloc((void**)&ctx->d_levels, sizeof(tr_level_table)) != hipSuccess ||
        hipMalloc((void**)&ctx->d_slice_thr, sizeof(float) * (TR_MAX_DEPTH_SLICES + 2)) != hipSuccess ||
        hipMalloc((void**)&ctx->d_colour_tables, sizeof(tr_colour_tables)) != hipSuccess) {
        (void)hipFree(ctx->d_front_ticket);
        (void)hipFree(ctx->d_levels);
        (void)hipFree(ctx->d_slice_thr);
        (void)hipFree(ctx->d_colour_tables);
        delete ctx;
        return TR_ERR_OUT_OF_MEMORY;
    }
    {
        tr_colour_tables tables;
        fill_colour_tables(&tables);
        if (hipMemcpy(ctx->d_colour_tables, &tables, sizeof(tables), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(ctx->d_levels);
            (void)hipFree(ctx->d_slice_thr);
            (void)hipFree(ctx->d_colour_tables);
            delete ctx;
            return TR_ERR_NO_DEVICE;
        }
    }
    *out_ctx = ctx;
    return TR_OK;
}

tr_status tr_context_destroy(tr_context* ctx) {
    if (!ctx) return TR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(ctx->d_materials_raw);
    (void)hipFree(ctx->d_dmats);
    (void)hipFree(ctx->d_dtaps);
    (void)hipFree(ctx->d_lights);
    (void)hipFree(ctx->d_alights);
    (void)hipFree(ctx->d_lut_rgba8);
    (void)hipFree(ctx->d_lut_pairs);
    (void)hipFree(ctx->d_lut_lines);
    (void)hipFree(ctx->d_levels);
    (void)hipFree(ctx->d_slice_thr);
    (void)hipFree(ctx->d_cluster_x);
    (void)hipFree(ctx->d_cluster_y_term);
    (void)hipFree(ctx->d_tex_arena);
    (void)hipFree(ctx->d_textures);
    (void)hipFree(ctx->d_colour_tables);
    (void)hipFree(ctx->d_front_ticket);
    free_geometry(ctx);
    (void)hipFree(ctx->d_vis[0]);
    if (ctx->tables_event) (void)hipEventDestroy(ctx->tables_event);
    delete ctx;
    return TR_OK;
}

tr_status tr_pyramid_layout(uint32_t width, uint32_t height, tr_pyramid* out, size_t* out_bytes) {
    if (!out || width == 0 || height == 0) return TR_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    out->width = width;
    out->height = height;
    uint32_t levels = mip_levels_for_size(width, height);
    if (levels > TR_MAX_MIP_LEVELS) levels = TR_MAX_MIP_LEVELS;
    out->levels = levels;
    uint64_t off = 0;
    for (uint32_t l = 0; l < levels; ++l) {
        out->level_offset[l] = (uint32_t)off;
        off += (uint64_t)level_dim(width, l) * level_dim(height, l);
    }
    if ((off + 1u) * 8u > 0xFFFFFFFFull) return TR_ERR_UNSUPPORTED;   // sampled with 32-bit byte offsets
    // + one texel of tail padding: the sampler reads texels in 16-byte pairs and may touch (never use) the 8
    // bytes after the last texel of the last level
    if (out_bytes) *out_bytes = (size_t)(off + 1u) * 8u;
    return TR_OK;
}

tr_status tr_upload_materials(tr_context* ctx, const tr_material_info* materials_host, uint32_t count, void* stream_) {
    if (!ctx || !materials_host || count == 0) return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    bool any_textured = false, any_full = false;
    uint32_t full_slots = 0;
    int32_t max_id = -1;
    for (uint32_t i = 0; i < count; ++i) {
        const tr_textures& t = materials_host[i].textures;
        // the slots the shaders read (shader/src/lib.rs:65-76, 120-124; lighting.rs:222-313); occlusion is never sampled
        const int32_t ids[8] = {t.diffuse, t.metallic_roughness, t.normal_map, t.emissive,
                                t.transmission, t.thickness, t.specular, t.specular_colour};
        bool textured = false, only_diffuse = true;
        uint32_t slots = 0;
        for (int k = 0; k < 8; ++k) {
            const int32_t id = ids[k];
            if (id < -1) return TR_ERR_INVALID_ARGUMENT;
            if (id != -1) {
                textured = true;
                slots |= 1u << k;
                if (k != 0) only_diffuse = false;
            }
            if (id > max_id) max_id = id;
        }
        any_textured |= textured;
        // the same rule as digest_materials_kernel's lite class (flags bit 3)
        const bool full = textured && !(only_diffuse && materials_host[i].metallic_factor == 0.0f);
        any_full |= full;
        if (full) full_slots |= slots;
    }
    ctx->any_textured = any_textured;
    ctx->any_full_textured = any_full;
    ctx->full_slots = full_slots;
    ctx->max_texture_id = max_id;
    if (count > ctx->cap_materials) {
        (void)hipFree(ctx->d_materials_raw);
        (void)hipFree(ctx->d_dmats);
        (void)hipFree(ctx->d_dtaps);
        ctx->d_materials_raw = nullptr;
        ctx->d_dmats = nullptr;
        ctx->d_dtaps = nullptr;
        ctx->cap_materials = 0;
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_materials_raw, sizeof(tr_material_info) * count));
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_dmats, sizeof(tr_dmat) * count));
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_dtaps, sizeof(tr_dtap) * count));
        ctx->cap_materials = count;
    }
    TR_TRY(tables_before_rebuild(ctx, stream));   // (launches of other streams may still read the previous records)
    ctx->stage_materials.assign(materials_host, materials_host + count);
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_materials_raw, ctx->stage_materials.data(), sizeof(tr_material_info) * count,
                               hipMemcpyHostToDevice, stream));
    ctx->num_materials = count;
    ctx->dmats_dirty = true;
    return tables_rebuilt(ctx, stream);
}

}  // extern "C"

// The digest of one light (tr_upload_lights / tr_update_lights): what the shading kernels and the cluster assignment read.
static void digest_light(const tr_light& s, tr_dlight& d, tr_alight& a) {
    std::memset(&d, 0, sizeof(d));
    std::memset(&a, 0, sizeof(a));
    for (int k = 0; k < 3; ++k) {
        d.pos[k] = s.position_and_spotlight_epsilon[k];
        d.colour[k] = s.colour_emission_and_falloff_distance_sq[k];
        d.spot_dir[k] = s.spotlight_direction_and_outer_angle[k];
        a.pos[k] = s.position_and_spotlight_epsilon[k];
        a.spot_dir[k] = s.spotlight_direction_and_outer_angle[k];
    }
    const float outer = s.spotlight_direction_and_outer_angle[3];
    d.is_spot = outer != 0.0f ? 1u : 0u;  // Light::is_a_spotlight, shared-structs/src/lib.rs:125-127
    d.cos_outer = std::cos(outer);
    d.inv_spot_epsilon = 1.0f / s.position_and_spotlight_epsilon[3];
    a.falloff_distance_sq = s.colour_emission_and_falloff_distance_sq[3];
    a.is_spot = d.is_spot;
    a.cos_angle = std::cos(outer);   // ClusterAabb::cull_spotlight, shared-structs/src/lib.rs:312
    a.sin_angle = std::sin(outer);
}

// Per-frame rewrites of a few records (the reference writes them into mapped buffers: two spotlights, src/main.rs:1244-1256;
// one instance's rotation, :1258-1261, 1316-1322): the records travel INSIDE the launch's kernel arguments, so the call
// neither allocates, nor copies from pageable host memory, nor waits — and is ordered on `stream` like any launch.
namespace tr {
struct alignas(16) tr_update_payload {   // (the records digested into it are 16-byte aligned types)
    uint32_t words[TR_UPDATE_MAX_BYTES / 4u];
};
__global__ __launch_bounds__(64) void update_records_kernel(uint32_t* __restrict__ dst_a, uint32_t words_a,
                                                            uint32_t* __restrict__ dst_b, uint32_t words_b, const tr_update_payload p) {
    // (payload: words_a words for dst_a, then words_b words for dst_b)
    for (uint32_t i = threadIdx.x; i < words_a + words_b; i += 64u) {
        if (i < words_a) dst_a[i] = p.words[i];
        else dst_b[i - words_a] = p.words[i];
    }
}
}  // namespace tr

extern "C" {

tr_status tr_upload_lights(tr_context* ctx, const tr_light* lights_host, uint32_t count, void* stream_) {
    if (!ctx || (!lights_host && count)) return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t alloc = count ? count : 1u;
    if (alloc > ctx->cap_lights) {
        (void)hipFree(ctx->d_lights);
        (void)hipFree(ctx->d_alights);
        ctx->d_lights = nullptr;
        ctx->d_alights = nullptr;
        ctx->cap_lights = 0;
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_lights, sizeof(tr_dlight) * alloc));
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_alights, sizeof(tr_alight) * alloc));
        ctx->cap_lights = alloc;
    }
    TR_TRY(tables_before_rebuild(ctx, stream));
    ctx->stage_lights.resize(alloc);
    ctx->stage_alights.resize(alloc);
    std::memset(ctx->stage_lights.data(), 0, sizeof(tr_dlight) * alloc);
    std::memset(ctx->stage_alights.data(), 0, sizeof(tr_alight) * alloc);
    for (uint32_t i = 0; i < count; ++i) digest_light(lights_host[i], ctx->stage_lights[i], ctx->stage_alights[i]);
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_lights, ctx->stage_lights.data(), sizeof(tr_dlight) * alloc,
                               hipMemcpyHostToDevice, stream));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_alights, ctx->stage_alights.data(), sizeof(tr_alight) * alloc,
                               hipMemcpyHostToDevice, stream));
    ctx->num_lights = count;
    return tables_rebuilt(ctx, stream);
}

tr_status tr_update_lights(tr_context* ctx, uint32_t first, uint32_t count, const tr_light* lights_host, void* stream_) {
    if (!ctx || (!lights_host && count)) return TR_ERR_INVALID_ARGUMENT;
    if ((uint64_t)first + count > ctx->num_lights) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0u) return TR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    constexpr uint32_t per_launch = TR_UPDATE_MAX_BYTES / (uint32_t)(sizeof(tr_dlight) + sizeof(tr_alight));
    for (uint32_t done = 0; done < count; done += per_launch) {
        const uint32_t n = std::min(per_launch, count - done);
        tr::tr_update_payload pay;
        tr_dlight* d = reinterpret_cast<tr_dlight*>(pay.words);
        tr_alight* a = reinterpret_cast<tr_alight*>(pay.words + n * (sizeof(tr_dlight) / 4u));
        for (uint32_t i = 0; i < n; ++i) digest_light(lights_host[done + i], d[i], a[i]);
        hipLaunchKernelGGL(tr::update_records_kernel, dim3(1), dim3(64), 0, stream,
                           reinterpret_cast<uint32_t*>(ctx->d_lights + first + done), n * (uint32_t)(sizeof(tr_dlight) / 4u),
                           reinterpret_cast<uint32_t*>(ctx->d_alights + first + done), n * (uint32_t)(sizeof(tr_alight) / 4u), pay);
    }
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_set_cluster_tables(tr_context* ctx, const void* counts_dev, const void* indices_dev,
                                uint32_t num_clusters_total) {
    if (!ctx || !counts_dev || !indices_dev || num_clusters_total == 0) return TR_ERR_INVALID_ARGUMENT;
    ctx->d_cluster_counts = (const uint32_t*)counts_dev;
    ctx->d_light_indices = (const uint32_t*)indices_dev;
    ctx->num_clusters_total = num_clusters_total;
    return TR_OK;
}

tr_status tr_upload_ggx_lut(tr_context* ctx, const uint8_t* rgba8_host, uint32_t width, uint32_t height,
                            void* stream_) {
    if (!ctx || !rgba8_host || width == 0 || height == 0) return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t stride = width + 2u;
    if (width != ctx->lut_w || height != ctx->lut_h) {
        (void)hipFree(ctx->d_lut_rgba8);
        (void)hipFree(ctx->d_lut_pairs);
        ctx->d_lut_rgba8 = ctx->d_lut_pairs = nullptr;
        ctx->lut_w = ctx->lut_h = ctx->lut_stride = 0;
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_lut_rgba8, (size_t)width * height * 4u));
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_lut_pairs, (size_t)stride * height * 4u));
    }
    TR_TRY(tables_before_rebuild(ctx, stream));
    ctx->stage_lut.assign(rgba8_host, rgba8_host + (size_t)width * height * 4u);
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_lut_rgba8, ctx->stage_lut.data(), ctx->stage_lut.size(), hipMemcpyHostToDevice,
                               stream));
    hipLaunchKernelGGL(build_lut_pairs_kernel, dim3((stride + 255) / 256, height), dim3(256), 0, stream,
                       ctx->d_lut_rgba8, ctx->d_lut_pairs, width, height, stride);
    TR_HIP(ctx, hipGetLastError());
    ctx->lut_w = width;
    ctx->lut_h = height;
    ctx->lut_stride = stride;
    ctx->dmats_dirty = ctx->num_materials > 0;  // LUT rows are part of the digested material
    return tables_rebuilt(ctx, stream);
}

tr_status tr_upload_textures(tr_context* ctx, const tr_texture_desc* textures_host, uint32_t count, void* stream_) {
    if (!ctx || (!textures_host && count)) return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<tr_dtex> table(count);
    uint64_t total = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const tr_texture_desc& d = textures_host[i];
        if (!d.rgba8 || d.width == 0 || d.height == 0) return TR_ERR_INVALID_ARGUMENT;
        tr_dtex& t = table[i];
        std::memset(&t, 0, sizeof(t));
        t.width = d.width;
        t.height = d.height;
        t.srgb = d.srgb ? 1u : 0u;
        t.levels = mip_levels_for_size(d.width, d.height);           // src/model_loading.rs:354
        if (t.levels > TR_MAX_MIP_LEVELS) t.levels = TR_MAX_MIP_LEVELS;
        t.wf = (float)d.width;
        t.hf = (float)d.height;
        t.max_lod = (float)(t.levels - 1u);
        total = (total + 63u) & ~63ull;                              // chains start on 256-byte boundaries
        for (uint32_t l = 0; l < TR_MAX_MIP_LEVELS + 4u; ++l) {
            if (l < t.levels) {
                t.offset[l] = (uint32_t)total;
                total += (uint64_t)level_dim(d.width, l) * level_dim(d.height, l);
            } else {
                t.offset[l] = t.offset[t.levels - 1u];
            }
        }
        if (total > 0xFFFFFFFFull) return TR_ERR_UNSUPPORTED;
    }
    // the previous array may still be read by work in flight on another stream
    TR_HIP(ctx, hipDeviceSynchronize());
    (void)hipFree(ctx->d_tex_arena);
    (void)hipFree(ctx->d_textures);
    ctx->d_tex_arena = nullptr;
    ctx->d_textures = nullptr;
    ctx->num_textures = 0;
    ctx->h_textures.clear();
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_tex_arena, (size_t)total * 4u));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_textures, sizeof(tr_dtex) * count));
    for (uint32_t i = 0; i < count; ++i) {
        const tr_dtex& t = table[i];
        TR_HIP(ctx, hipMemcpyAsync(ctx->d_tex_arena + t.offset[0], textures_host[i].rgba8, (size_t)t.width * t.height * 4u,
                                   hipMemcpyHostToDevice, stream));
        for (uint32_t l = 1; l < t.levels; ++l) {
            const uint32_t ws = level_dim(t.width, l - 1), hs = level_dim(t.height, l - 1);
            const uint32_t wd = level_dim(t.width, l), hd = level_dim(t.height, l);
            hipLaunchKernelGGL(texture_downsample_kernel, dim3((wd + 63u) / 64u, (hd + 3u) / 4u), dim3(256), 0, stream,
                               (const uint32_t*)(ctx->d_tex_arena + t.offset[l - 1]), ctx->d_tex_arena + t.offset[l], ws, hs,
                               wd, hd, t.srgb, (const tr_colour_tables*)ctx->d_colour_tables);
        }
    }
    TR_HIP(ctx, hipGetLastError());
    ctx->h_textures = table;
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_textures, ctx->h_textures.data(), sizeof(tr_dtex) * count, hipMemcpyHostToDevice,
                               stream));
    // the host images are caller memory: do not return before the copies have read them
    TR_HIP(ctx, hipStreamSynchronize(stream));
    ctx->num_textures = count;
    return TR_OK;
}

tr_status tr_texture_get_layout(const tr_context* ctx, uint32_t index, tr_texture_layout* out) {
    if (!ctx || !out || index >= ctx->num_textures) return TR_ERR_INVALID_ARGUMENT;
    const tr_dtex& t = ctx->h_textures[index];
    std::memset(out, 0, sizeof(*out));
    out->width = t.width;
    out->height = t.height;
    out->levels = t.levels;
    out->srgb = t.srgb;
    uint32_t total = 0;
    for (uint32_t l = 0; l < t.levels; ++l) {
        out->level_offset[l] = t.offset[l] - t.offset[0];
        total += level_dim(t.width, l) * level_dim(t.height, l);
    }
    out->total_texels = total;
    return TR_OK;
}

tr_status tr_download_texture(tr_context* ctx, uint32_t index, void* rgba8_host_out, size_t capacity_bytes, void* stream_) {
    if (!ctx || !rgba8_host_out || index >= ctx->num_textures) return TR_ERR_INVALID_ARGUMENT;
    tr_texture_layout lay;
    tr_status st = tr_texture_get_layout(ctx, index, &lay);
    if (st != TR_OK) return st;
    if (capacity_bytes < (size_t)lay.total_texels * 4u) return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    TR_HIP(ctx, hipMemcpyAsync(rgba8_host_out, ctx->d_tex_arena + ctx->h_textures[index].offset[0],
                               (size_t)lay.total_texels * 4u, hipMemcpyDeviceToHost, stream));
    TR_HIP(ctx, hipStreamSynchronize(stream));
    return TR_OK;
}

tr_status tr_frustum_culling(tr_context* ctx, const void* primitives, uint32_t num_primitives, const void* instances,
                             uint32_t num_instances, const tr_culling_push_constants* push, void* instance_counts,
                             void* stream_) {
    if (!ctx || !primitives || !push || !instance_counts || num_primitives == 0 || (!instances && num_instances))
        return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    TR_HIP(ctx, zero_fill(instance_counts, sizeof(uint32_t) * num_primitives, stream));   // src/main.rs:1668-1674
    if (instance_counts == ctx->d_instance_counts) ctx->counts_clean = false;   // (left holding this frame's counts)
    if (num_instances == 0) return TR_OK;
    tr_cull_params p;
    p.pc = *push;
    p.num_instances = num_instances;
    p.num_primitives = num_primitives;
    hipLaunchKernelGGL(frustum_culling_kernel, dim3((num_instances + 255u) / 256u), dim3(256), 0, stream, p,
                       (const tr_primitive_info*)primitives, (const tr_instance*)instances, (uint32_t*)instance_counts);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_demultiplex_draws(tr_context* ctx, const void* primitives, uint32_t num_primitives,
                               const void* instance_counts, void* draw_counts, void* const draws[TR_NUM_DRAW_BUFFERS],
                               void* stream_) {
    if (!ctx || !primitives || !instance_counts || !draw_counts || !draws || num_primitives == 0)
        return TR_ERR_INVALID_ARGUMENT;
    tr_draw_buffers out;
    for (uint32_t k = 0; k < TR_NUM_DRAW_BUFFERS; ++k) {
        if (!draws[k]) return TR_ERR_INVALID_ARGUMENT;
        out.draws[k] = (tr_draw_command*)draws[k];
    }
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(demultiplex_draws_kernel, dim3(1), dim3(1024), 0, stream, (const tr_primitive_info*)primitives,
                       (const uint32_t*)instance_counts, num_primitives, (uint32_t*)draw_counts, out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_upload_geometry(tr_context* ctx, const tr_geometry_desc* g, void* stream_) {
    if (!ctx || !g || !g->position || !g->normal || !g->uv || !g->index || !g->primitives || !g->instances ||
        g->num_vertices == 0 || g->num_indices == 0 || g->num_primitives == 0 || g->num_instances == 0)
        return TR_ERR_INVALID_ARGUMENT;
    // every index / id is checked once here, so the kernels can index unchecked like the reference's shaders
    for (uint32_t i = 0; i < g->num_indices; ++i)
        if (g->index[i] >= g->num_vertices) return TR_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> per_primitive(g->num_primitives, 0u);
    for (uint32_t i = 0; i < g->num_instances; ++i) {
        if (g->instances[i].primitive_id >= g->num_primitives) return TR_ERR_INVALID_ARGUMENT;
        per_primitive[g->instances[i].primitive_id] += 1u;
    }
    uint64_t max_tris[2] = {0, 0};
    for (uint32_t p = 0; p < g->num_primitives; ++p) {
        const tr_primitive_info& pi = g->primitives[p];
        if ((uint64_t)pi.first_index + pi.index_count > g->num_indices) return TR_ERR_INVALID_ARGUMENT;
        if ((uint64_t)pi.first_instance + per_primitive[p] > g->num_instances) return TR_ERR_INVALID_ARGUMENT;
        max_tris[pi.draw_buffer_index >= 2u ? 1 : 0] += (uint64_t)(pi.index_count / 3u) * per_primitive[p];
    }
    if (max_tris[0] > 0xFFFFFFF0ull || max_tris[1] > 0xFFFFFFF0ull) return TR_ERR_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    TR_HIP(ctx, hipDeviceSynchronize());
    free_geometry(ctx);
    const size_t nv = g->num_vertices, ni = g->num_indices, np_ = g->num_primitives, nn = g->num_instances;
    const size_t cap = (size_t)std::max<uint64_t>(std::max(max_tris[0], max_tris[1]), 1ull);
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_position, nv * 12u));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_normal, nv * 12u));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_uv, nv * 8u));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_index, ni * 4u));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_primitives, np_ * sizeof(tr_primitive_info)));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_instances, nn * sizeof(tr_instance)));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_instance_counts, np_ * 4u));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_draw_counts, TR_NUM_DRAW_BUFFERS * 4u));
    for (auto& d : ctx->d_draws) TR_HIP(ctx, hipMalloc((void**)&d, np_ * sizeof(tr_draw_command)));
    // rasteriser work buffers, one set per layer (the two front ends run in the same launches)
    ctx->work_capacity = cap;
    ctx->work_draws = np_ + 1u;
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_tri_base, 2u * (np_ + 1u) * 4u));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_records, 2u * cap * sizeof(tr_tri_record)));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_tri_planes, 2u * cap * sizeof(tr_tri_planes)));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_item_base, 2u * (cap + 1u) * 4u));
    ctx->scan_blocks = (uint32_t)((cap + 255u) / 256u);
    // (behind the status words: the frame counter they are tagged with — on the device, advanced by every rasteriser launch,
    //  so that a captured frame replays with a new tag each time; a tag of 0 = never written)
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_scan_status, (2u * (size_t)ctx->scan_blocks + 1u) * 8u));
    TR_HIP(ctx, hipMemsetAsync(ctx->d_scan_status, 0, (2u * (size_t)ctx->scan_blocks + 1u) * 8u, stream));
    TR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctx->d_scan_status + 2u * (size_t)ctx->scan_blocks), 1, 1, stream));
    TR_HIP(ctx, hipMalloc((void**)&ctx->d_layer_counts, 2u * sizeof(tr_layer_counts)));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_position, g->position, nv * 12u, hipMemcpyHostToDevice, stream));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_normal, g->normal, nv * 12u, hipMemcpyHostToDevice, stream));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_uv, g->uv, nv * 8u, hipMemcpyHostToDevice, stream));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_index, g->index, ni * 4u, hipMemcpyHostToDevice, stream));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_primitives, g->primitives, np_ * sizeof(tr_primitive_info), hipMemcpyHostToDevice, stream));
    TR_HIP(ctx, hipMemcpyAsync(ctx->d_instances, g->instances, nn * sizeof(tr_instance), hipMemcpyHostToDevice, stream));
    TR_HIP(ctx, hipStreamSynchronize(stream));   // the host arrays are caller memory
    ctx->num_vertices = g->num_vertices;
    ctx->num_indices = g->num_indices;
    ctx->num_primitives = g->num_primitives;
    ctx->num_instances = g->num_instances;
    ctx->max_triangles[0] = (uint32_t)max_tris[0];
    ctx->max_triangles[1] = (uint32_t)max_tris[1];
    ctx->counts_clean = false;
    ctx->h_instance_primitive.resize(nn);
    for (size_t i = 0; i < nn; ++i) ctx->h_instance_primitive[i] = g->instances[i].primitive_id;
    return TR_OK;
}

tr_status tr_update_instances(tr_context* ctx, uint32_t first, uint32_t count, const tr_instance* instances_host, void* stream_) {
    if (!ctx || (!instances_host && count)) return TR_ERR_INVALID_ARGUMENT;
    if ((uint64_t)first + count > ctx->num_instances) return TR_ERR_INVALID_ARGUMENT;
    // (an instance keeps its primitive: the per-primitive instance ranges, the draw streams and the rasteriser's work buffers
    //  were sized from them at tr_upload_geometry — what changes per frame is the transform, src/main.rs:1258-1261)
    for (uint32_t i = 0; i < count; ++i)
        if (instances_host[i].primitive_id != ctx->h_instance_primitive[first + i]) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0u) return TR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    constexpr uint32_t per_launch = TR_UPDATE_MAX_BYTES / (uint32_t)sizeof(tr_instance);
    for (uint32_t done = 0; done < count; done += per_launch) {
        const uint32_t n = std::min(per_launch, count - done);
        tr::tr_update_payload pay;
        std::memcpy(pay.words, instances_host + done, (size_t)n * sizeof(tr_instance));
        hipLaunchKernelGGL(tr::update_records_kernel, dim3(1), dim3(64), 0, stream,
                           reinterpret_cast<uint32_t*>(ctx->d_instances + first + done), n * (uint32_t)(sizeof(tr_instance) / 4u),
                           (uint32_t*)nullptr, 0u, pay);
    }
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

}  // extern "C"

namespace {
// The entries that rasterise can be captured into a HIP graph once a frame of the same size has run outside a capture:
// what they refuse under capture is what cannot be captured — growing the visibility buffers (ensure_vis_buffers) and
// rebuilding a table (tables_before_rebuild).
bool stream_is_capturing(void* stream) {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing((hipStream_t)stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone;
}

// The rasteriser's per-frame-size buffers: both layers' visibility words, behind them the two tile coverage maps.
tr_status ensure_vis_buffers(tr_context* ctx, uint32_t w, uint32_t h, void* stream) {
    const size_t npix = (size_t)w * h;
    if ((npix > ctx->vis_pixels || ctx->vis_w != w || ctx->vis_h != h) && stream_is_capturing(stream)) return TR_ERR_UNSUPPORTED;
    if (npix > ctx->vis_pixels) {
        TR_HIP(ctx, hipDeviceSynchronize());
        (void)hipFree(ctx->d_vis[0]);
        ctx->d_vis[0] = ctx->d_vis[1] = nullptr;
        ctx->vis_pixels = 0;
        ctx->d_tile_cover[0] = ctx->d_tile_cover[1] = nullptr;
        TR_HIP(ctx, hipMalloc((void**)&ctx->d_vis[0], 2u * npix * 8u + 3u * (npix / 64u + 65536u + 16384u) * 4u + (kFrontLists * 65u + 64u) * 4u));
        ctx->vis_pixels = npix;
        ctx->vis_clean = false;
    }
    if (ctx->vis_w != w || ctx->vis_h != h) {   // another frame size lays the buffers out differently: the previous
        ctx->vis_clean = false;                  // size's coverage words may lie where this size's visibility words do
        ctx->vis_w = w;
        ctx->vis_h = h;
    }
    ctx->d_vis[1] = ctx->d_vis[0] + npix;
    const size_t cover_tiles = (size_t)((w + 63u) / 64u) * ((h + 3u) / 4u);
    ctx->d_tile_cover[0] = (uint32_t*)(ctx->d_vis[0] + 2u * npix);
    ctx->d_tile_cover[1] = ctx->d_tile_cover[0] + cover_tiles;
    ctx->d_front_list_count = ctx->d_tile_cover[1] + cover_tiles;   // (zeroed with the maps: cover_clear_bytes)
    ctx->d_front_list = ctx->d_front_list_count + kFrontLists;
    ctx->front_list_cap = (uint32_t)(cover_tiles / kFrontLists + 1u);   // (tile t is listed in sub-list t % kFrontLists)
    return TR_OK;
}
// what a frame zeroes before rasterising: the maps and the counter of the list of transmissive-covered tiles behind them,
// rounded up to whole 16-byte vectors (the round-up reaches into the list, which is rebuilt every frame)
inline size_t cover_clear_bytes(uint32_t w, uint32_t h) {
    const size_t cover_tiles = (size_t)((w + 63u) / 64u) * ((h + 3u) / 4u);
    return ((2u * cover_tiles + kFrontLists) * 4u + 15u) & ~(size_t)15u;
}

// The work buffers of both layers as the front-end kernels take them (after ensure_vis_buffers).
void fill_two_layers(const tr_context* ctx, const void* const draws[TR_NUM_DRAW_BUFFERS], const tr_gbuffer_target* const targets[2],
                     tr_two_layers& two) {
    const size_t cap = ctx->work_capacity;
    for (uint32_t layer = 0; layer < 2u; ++layer) {
        tr_layer_work& W = two.l[layer];
        W.draws_a = (const tr_draw_command*)draws[layer * 2u];
        W.draws_b = (const tr_draw_command*)draws[layer * 2u + 1u];
        W.buffer_a = layer * 2u;
        W.capacity_triangles = ctx->max_triangles[layer];
        W.tri_base = ctx->d_tri_base + layer * ctx->work_draws;
        W.counts = ctx->d_layer_counts + layer;
        W.records = ctx->d_records + layer * cap;
        W.tri_planes = ctx->d_tri_planes + layer * cap;
        W.item_base = ctx->d_item_base + layer * (cap + 1u);
        W.vis = ctx->d_vis[layer];
        W.planes.pos_depth = (float4*)targets[layer]->pos_depth;
        W.planes.nrm_scale = (float4*)targets[layer]->nrm_scale;
        W.planes.uv = (float2*)targets[layer]->uv;
        W.planes.material_id = (uint32_t*)targets[layer]->material_id;
        W.tile_cover = ctx->d_tile_cover[layer];
    }
}


// fused_demux: the frame recorder's call — its first launch has demultiplexed the draws and scanned both layers' draw
// streams behind its culling blocks (frame_front_kernel) and zeroed the coverage maps.
// resolve: write the TGB-v1 planes.  Without it the layers stay visibility words + triangle planes, which the caller
// hands to VIS shading launches (they zero the words; the caller sets ctx->vis_clean once both passes are enqueued).
tr_status rasterize_impl(tr_context* ctx, const void* draw_counts, const void* const draws[TR_NUM_DRAW_BUFFERS],
                         const tr_push_constants* push, const tr_gbuffer_target* opaque,
                         const tr_gbuffer_target* transmissive, void* stream_, bool fused_demux, bool resolve) {
    if (!ctx || !draw_counts || !draws || !push || !opaque || !transmissive) return TR_ERR_INVALID_ARGUMENT;
    for (uint32_t k = 0; k < TR_NUM_DRAW_BUFFERS; ++k)
        if (!draws[k]) return TR_ERR_INVALID_ARGUMENT;
    const tr_gbuffer_target* targets[2] = {opaque, transmissive};
    for (const tr_gbuffer_target* t : targets)
        if (!t->pos_depth || !t->nrm_scale || !t->uv || !t->material_id) return TR_ERR_INVALID_ARGUMENT;
    const uint32_t w = push->framebuffer_size[0], h = push->framebuffer_size[1];
    if (w == 0 || h == 0 || w > 65535u || h > 65535u) return TR_ERR_INVALID_ARGUMENT;
    if (!ctx->d_position || ctx->num_materials == 0) return TR_ERR_TABLES_MISSING;
    if (ctx->max_texture_id >= (int32_t)ctx->num_textures) return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)w * h;
    {
        const tr_status vs = ensure_vis_buffers(ctx, w, h, stream_);
        if (vs != TR_OK) return vs;
    }
    tr_geometry_view gv;
    gv.position = ctx->d_position;
    gv.normal = ctx->d_normal;
    gv.uv = ctx->d_uv;
    gv.index = ctx->d_index;
    gv.instances = ctx->d_instances;
    tr_raster_frame fr;
    std::memcpy(fr.proj_view, push->proj_view, sizeof(fr.proj_view));
    fr.width = w;
    fr.height = h;
    tr_alpha_tables at;
    at.materials = ctx->d_materials_raw;
    at.textures = ctx->d_textures;
    at.tex_arena = ctx->d_tex_arena;
    at.num_textures = ctx->num_textures;
    tr_two_layers two;
    fill_two_layers(ctx, draws, targets, two);
    // The visibility buffers are zero on entry: filled once after (re)allocation, and every resolve zeroes the words its
    // frame set (raster_resolve_body).  Per frame only the two tile coverage maps are cleared (260 KB at 4K).
    // (the flag describes the context when the call is ENQUEUED — for a captured frame: at capture time.  133 MB at 4K is
    //  not a clear to record unconditionally; the header says what a replay must not follow)
    if (!ctx->vis_clean) TR_HIP(ctx, zero_fill(ctx->d_vis[0], 2u * npix * 8u, stream));
    ctx->vis_clean = false;   // (until this frame's resolve is enqueued)
    if (!(fused_demux && ctx->cover_cleared))   // (the frame recorder's first launch has zeroed them for its own call)
        TR_HIP(ctx, zero_fill(ctx->d_tile_cover[0], cover_clear_bytes(w, h), stream));
    ctx->cover_cleared = false;
    const uint32_t max_cap = std::max(ctx->max_triangles[0], ctx->max_triangles[1]);
    // (the set-up tags every triangle with its material's class for the tile coverage words the shading launches steer by:
    //  needs the digested material table; without a GGX LUT there is none yet and every triangle is tagged with both classes)
    const uint32_t* mat_flags = nullptr;
    if (ensure_digested(ctx, stream) == TR_OK)
        mat_flags = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ctx->d_dmats) + offsetof(tr_dmat, flags));
    TR_TRY(tables_acquire(ctx, stream));
    if (max_cap > 0u) {
        if (!fused_demux)
            hipLaunchKernelGGL(raster_scan_draws_kernel, dim3(1, 2), dim3(1024), 0, stream, two, (const uint32_t*)draw_counts,
                               ctx->num_primitives);
        // set-up + the work-item prefix (decoupled look-back; its status words are tagged with the frame counter the device
        // keeps behind them, which the rasteriser launch that follows advances)
        uint32_t* const epoch_word = reinterpret_cast<uint32_t*>(ctx->d_scan_status + 2u * (size_t)ctx->scan_blocks);
        hipLaunchKernelGGL(raster_setup_kernel, dim3((max_cap + 255u) / 256u, 2), dim3(256), 0, stream, gv, fr, two, mat_flags,
                           (uint32_t)(sizeof(tr_dmat) / 4u), ctx->d_scan_status, ctx->scan_blocks, (const uint32_t*)epoch_word);
        tr_raster_layers rl;
        for (uint32_t layer = 0; layer < 2u; ++layer) {
            const tr_layer_work& W = two.l[layer];
            rl.records[layer] = W.records;
            rl.item_base[layer] = W.item_base;
            rl.counts[layer] = W.counts;
            rl.vis[layer] = ctx->d_vis[layer];
            rl.tile_cover[layer] = ctx->d_tile_cover[layer];
            rl.enabled[layer] = ctx->max_triangles[layer] != 0u ? 1u : 0u;
        }
        // (a wave prepares its items — search, record, block tests — whether it has one or twenty: a small frame's fewer items
        //  go to fewer waves.  1080p mesh frame, 3 / 4 / 5 / 6 workgroups per CU: 86.3 / 85.5 / 88.3 / 88.7 us; 720p and 1440p flat)
        const uint32_t raster_wgs = npix < 3000000u ? std::min(ctx->raster_wgs_per_cu, 4u) : ctx->raster_wgs_per_cu;
        hipLaunchKernelGGL(raster_kernel, dim3(ctx->num_cus * raster_wgs, 2), dim3(256), 0, stream, gv, fr, rl, at, epoch_word);
    }
    if (resolve) {
        hipLaunchKernelGGL(raster_resolve_kernel, dim3((w + 63u) / 64u, (h + 3u) / 4u), dim3(256), 0, stream, fr, two,
                           fused_demux ? 0u : 1u);   // (fused = the frame recorder's call: its shading skips untouched tiles)
        ctx->vis_clean = true;
    }
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}
}  // namespace

extern "C" {

tr_status tr_rasterize(tr_context* ctx, const void* draw_counts, const void* const draws[TR_NUM_DRAW_BUFFERS],
                       const tr_push_constants* push, const tr_gbuffer_target* opaque,
                       const tr_gbuffer_target* transmissive, void* stream) {
    return rasterize_impl(ctx, draw_counts, draws, push, opaque, transmissive, stream, false, true);
}

tr_status tr_draw_scene(tr_context* ctx, const tr_culling_push_constants* culling, const tr_push_constants* push,
                        const tr_gbuffer_target* opaque, const tr_gbuffer_target* transmissive, void* stream) {
    if (!ctx || !culling || !push) return TR_ERR_INVALID_ARGUMENT;
    if (!ctx->d_position) return TR_ERR_TABLES_MISSING;
    tr_status st = tr_frustum_culling(ctx, ctx->d_primitives, ctx->num_primitives, ctx->d_instances, ctx->num_instances,
                                      culling, ctx->d_instance_counts, stream);
    if (st != TR_OK) return st;
    void* draws[TR_NUM_DRAW_BUFFERS] = {ctx->d_draws[0], ctx->d_draws[1], ctx->d_draws[2], ctx->d_draws[3]};
    st = tr_demultiplex_draws(ctx, ctx->d_primitives, ctx->num_primitives, ctx->d_instance_counts, ctx->d_draw_counts, draws,
                              stream);
    if (st != TR_OK) return st;
    return tr_rasterize(ctx, ctx->d_draw_counts, draws, push, opaque, transmissive, stream);
}

tr_status tr_write_cluster_data(tr_context* ctx, const tr_uniforms* u, const float inverse_perspective[16],
                                const uint32_t screen_dimensions[2], void* cluster_aabbs_out, void* stream_) {
    if (!ctx || !u || !inverse_perspective || !screen_dimensions || !cluster_aabbs_out) return TR_ERR_INVALID_ARGUMENT;
    const tr_light_cluster_coefficients& c = u->light_clustering_coefficients;
    if (c.num_depth_slices == 0 || c.num_depth_slices > TR_MAX_DEPTH_SLICES || u->num_clusters[0] == 0 ||
        u->num_clusters[1] == 0)
        return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    tr_cluster_build_params p;
    std::memset(&p, 0, sizeof(p));
    std::memcpy(p.inverse_perspective, inverse_perspective, sizeof(p.inverse_perspective));
    p.cluster_size_px[0] = u->cluster_size_in_pixels[0];
    p.cluster_size_px[1] = u->cluster_size_in_pixels[1];
    p.screen_dims[0] = (float)screen_dimensions[0];
    p.screen_dims[1] = (float)screen_dimensions[1];
    p.nx = u->num_clusters[0];
    p.ny = u->num_clusters[1];
    p.nz = c.num_depth_slices;
    for (uint32_t s = 0; s <= p.nz; ++s)   // slice_to_depth, shared-structs/src/lib.rs:65-67 (libm powf, like the oracle)
        p.slice_depth[s] = -c.z_near * std::pow(c.z_far / c.z_near, (float)s / (float)c.num_depth_slices);
    const uint32_t total = p.nx * p.ny * p.nz;
    hipLaunchKernelGGL(write_cluster_data_kernel, dim3((total + 63u) / 64u), dim3(64), 0, stream, p,
                       (tr_cluster_aabb*)cluster_aabbs_out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_assign_lights_to_clusters(tr_context* ctx, const float view_matrix[16], const float view_rotation[4],
                                       const void* cluster_aabbs, uint32_t num_clusters, void* counts_out,
                                       void* indices_out, void* stream_) {
    if (!ctx || !view_matrix || !view_rotation || !cluster_aabbs || !counts_out || !indices_out || num_clusters == 0)
        return TR_ERR_INVALID_ARGUMENT;
    if (!ctx->d_alights) return TR_ERR_TABLES_MISSING;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    tr_assign_params p;
    std::memcpy(p.view_matrix, view_matrix, sizeof(p.view_matrix));
    std::memcpy(p.view_rotation, view_rotation, sizeof(p.view_rotation));
    p.num_lights = ctx->num_lights;
    p.num_clusters = num_clusters;
    hipLaunchKernelGGL(assign_lights_kernel, dim3((num_clusters + 3u) / 4u), dim3(256), 0, stream, p, ctx->d_alights,
                       (const tr_cluster_aabb*)cluster_aabbs, (uint32_t*)counts_out, (uint32_t*)indices_out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_depth_slice_thresholds(const tr_light_cluster_coefficients* c, float* thresholds_out, uint32_t* max_slice_out) {
    if (!c || !thresholds_out || !max_slice_out) return TR_ERR_INVALID_ARGUMENT;
    return build_slice_thresholds(*c, thresholds_out, max_slice_out);
}

tr_status tr_get_depth_slice(tr_context* ctx, const tr_light_cluster_coefficients* c, const void* frag_depth, uint32_t count,
                             void* slices_out, void* stream_) {
    if (!ctx || !c || !frag_depth || !slices_out || ((uintptr_t)frag_depth & 3u) || ((uintptr_t)slices_out & 3u))
        return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    tr_uniforms u;
    std::memset(&u, 0, sizeof(u));
    u.light_clustering_coefficients = *c;
    tr_status st = ensure_slice_thresholds(ctx, &u, stream);
    if (st != TR_OK) return st;
    st = tables_acquire(ctx, stream);
    if (st != TR_OK) return st;
    slice_params sp;
    sp.scale = c->scale;
    sp.fpn = c->z_far + c->z_near;
    sp.fmn = c->z_far - c->z_near;
    sp.k = (float)(std::log2(2.0 * (double)c->z_near * (double)c->z_far) * (double)c->scale + (double)c->bias);
    sp.max = ctx->slice_max;
    sp.thr = ctx->d_slice_thr;
    hipLaunchKernelGGL(depth_slice_kernel, dim3((count + 255u) / 256u), dim3(256), 0, stream, (const float*)frag_depth, count,
                       sp, (uint32_t*)slices_out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

}   // extern "C"
namespace {
// tr_shade_opaque / tr_shade_opaque_pyramid.  `mip1_out`: level 1 of the opaque pyramid, written by the launch from the 2x2
// quads of the values it stores (tr_kernels.h: the pass's wave tiles hold whole quads) — the caller has checked that the
// frame sizes are even and the rect lies on even pixels; NULL: level 0 only.
tr_status shade_opaque_impl(tr_context* ctx, const tr_gbuffer* g, const tr_uniforms* u, const tr_push_constants* pc,
                            void* hdr_out, tr_format format, void* opaque_mip0_out, void* mip1_out, tr_rect rect, void* stream_) {
    if (!ctx || !g || !u || !pc || !hdr_out || !g->pos_depth || !g->nrm_scale || !g->material_id)
        return TR_ERR_INVALID_ARGUMENT;
    if (format != TR_FORMAT_RGBA16F && format != TR_FORMAT_RGBA32F) return TR_ERR_INVALID_ARGUMENT;
    if (!tables_ready(ctx, false)) return TR_ERR_TABLES_MISSING;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    tr_frame_params fp;
    tr_status st = ensure_slice_thresholds(ctx, u, stream);
    if (st != TR_OK) return st;
    st = fill_frame_params(ctx, g, u, pc, rect, &fp);
    if (st != TR_OK) return st;
    st = ensure_digested(ctx, stream);
    if (st != TR_OK) return st;
    st = ensure_cluster_tables(ctx, u, fp.width, fp.height, stream);
    if (st != TR_OK) return st;
    st = check_textured_launch(ctx, g, fp);
    if (st != TR_OK) return st;
    st = tables_acquire(ctx, stream);
    if (st != TR_OK) return st;
    fp.pyr_levels = 1;
    // one-wave workgroups: blocks_per_xcd counts units of four waves
    tr_launch L;
    fill_launch(L, ctx, fp, g);
    const dim3 grid(persistent_grid(ctx, fp.tiles_x * fp.tiles_y, L.vis != nullptr) * 4u), block(64);
    {
        L.hdr = hdr_out;
        L.mip0 = (uint2*)opaque_mip0_out;
        L.mip1 = (L.vis && L.mip0 && format == TR_FORMAT_RGBA16F) ? (uint2*)ctx->mip1_hint : nullptr;   // (the frame recorder)
        if (!L.vis && L.mip0 && format == TR_FORMAT_RGBA16F && mip1_out) L.mip1 = (uint2*)mip1_out;     // (tr_shade_opaque_pyramid)
        if (L.vis && L.cover_front && ctx->front_list_hint) {   // ... whose opaque launch lists the tiles with transmissive fragments
            L.front_list_build = ctx->d_front_list;
            L.front_list_build_count = ctx->d_front_list_count;
            L.front_list_cap = ctx->front_list_cap;
        }
        const bool half = format == TR_FORMAT_RGBA16F;
        if (ctx->any_textured) {
            launch_textured<false>(ctx, L, half, grid, block, stream);
        } else {
            launch_shade<false, kTexNone>(L, half, grid, block, stream);
        }
    }
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}
}  // namespace
extern "C" {

tr_status tr_shade_opaque(tr_context* ctx, const tr_gbuffer* g, const tr_uniforms* u, const tr_push_constants* pc,
                          void* hdr_out, tr_format format, void* opaque_mip0_out, tr_rect rect, void* stream) {
    return shade_opaque_impl(ctx, g, u, pc, hdr_out, format, opaque_mip0_out, nullptr, rect, stream);
}

tr_status tr_shade_opaque_pyramid(tr_context* ctx, const tr_gbuffer* g, const tr_uniforms* u, const tr_push_constants* pc,
                                  void* hdr_out, tr_format format, const tr_pyramid* p, tr_rect rect,
                                  uint32_t* next_level_out, void* stream) {
    if (!p || !p->texels || !next_level_out || !pc || p->levels == 0 || p->levels > TR_MAX_MIP_LEVELS) return TR_ERR_INVALID_ARGUMENT;
    if (p->width != pc->framebuffer_size[0] || p->height != pc->framebuffer_size[1]) return TR_ERR_INVALID_ARGUMENT;
    uint2* const base = (uint2*)p->texels;
    // level 1 from the pass's own quads: an exact 2x2 box needs even frame sizes, and whole quads a rect on even pixels —
    // otherwise the chain starts at level 1 as after tr_shade_opaque
    const bool quads = p->levels >= 2u && format == TR_FORMAT_RGBA16F && !(p->width & 1u) && !(p->height & 1u) &&
                       !((rect.x0 | rect.y0 | rect.x1 | rect.y1) & 1u);
    *next_level_out = quads ? 2u : 1u;
    return shade_opaque_impl(ctx, g, u, pc, hdr_out, format, base + p->level_offset[0], quads ? base + p->level_offset[1] : nullptr, rect, stream);
}

}   // extern "C"
namespace {
// Levels first .. levels - 1 of the pyramid from level first - 1 (tr_generate_mips: first = 1; the frame recorder, whose
// opaque launches have written level 1 themselves: first = 2).
tr_status generate_mips_from(tr_context* ctx, const tr_pyramid* p, uint32_t first, void* stream_) {
    if (!ctx || !p || !p->texels || p->levels == 0 || p->levels > TR_MAX_MIP_LEVELS) return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    uint2* base = (uint2*)p->texels;
    uint32_t l = first;
    // (1) leading levels whose parent has even sizes: exact 2x2 boxes, up to five levels per launch from LDS
    while (l < p->levels) {
        const uint32_t ws = level_dim(p->width, l - 1), hs = level_dim(p->height, l - 1);
        uint32_t n = 0;
        for (uint32_t k = 0; k < 5u && l + k < p->levels; ++k) {
            const uint32_t wk = ws >> k, hk = hs >> k;
            if (wk < 2u || hk < 2u || (wk & 1u) || (hk & 1u)) break;
            ++n;
        }
        if (n == 0) break;
        tr_mip_even_params ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.w0 = ws;
        ep.h0 = hs;
        ep.nlevels = n;
        ep.src_offset = p->level_offset[l - 1];
        for (uint32_t k = 0; k < n; ++k) ep.dst_offset[k] = p->level_offset[l + k];
        ep.row_begin = 0u;
        ep.row_end = hs >> 1;
        hipLaunchKernelGGL(mip_even_kernel, dim3(((ws >> 1) + 15u) / 16u, ((hs >> 1) + 15u) / 16u), dim3(256), 0, stream,
                           base, ep);
        l += n;
    }
    // (2) levels with an odd parent that are still too large for one workgroup: one general LINEAR blit each
    while (l < p->levels && (uint64_t)level_dim(p->width, l) * level_dim(p->height, l) > kMipTailMaxTexels) {
        const uint32_t ws = level_dim(p->width, l - 1), hs = level_dim(p->height, l - 1);
        const uint32_t wd = level_dim(p->width, l), hd = level_dim(p->height, l);
        hipLaunchKernelGGL(downsample_kernel, dim3((wd + 63) / 64, (hd + 3) / 4), dim3(256), 0, stream,
                           (const uint2*)(base + p->level_offset[l - 1]), base + p->level_offset[l], ws, hs, wd, hd);
        ++l;
    }
    // (3) the tail, one workgroup, previous level in LDS
    if (l < p->levels) {
        tr_mip_tail_params tp;
        std::memset(&tp, 0, sizeof(tp));
        tp.first = l;
        tp.levels = p->levels;
        size_t lds_texels = 0;
        for (uint32_t k = 0; k < p->levels; ++k) {
            tp.offset[k] = p->level_offset[k];
            tp.width[k] = level_dim(p->width, k);
            tp.height[k] = level_dim(p->height, k);
            if (k >= l) lds_texels += (size_t)tp.width[k] * tp.height[k];
        }
        if (!ctx->mip_tail_attr_set) {   // function attributes are per device: kept in the context, not process-wide
            TR_HIP(ctx, hipFuncSetAttribute((const void*)mip_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->mip_tail_attr_set = true;
        }
        hipLaunchKernelGGL(mip_tail_kernel, dim3(1), dim3(1024), lds_texels * sizeof(uint2), stream, base, tp);
    }
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}
}  // namespace
extern "C" {

tr_status tr_generate_mips(tr_context* ctx, const tr_pyramid* p, void* stream) { return generate_mips_from(ctx, p, 1u, stream); }

tr_status tr_generate_mips_from(tr_context* ctx, const tr_pyramid* p, uint32_t first_level, void* stream) {
    if (first_level == 0u) return TR_ERR_INVALID_ARGUMENT;
    if (p && first_level >= p->levels) return TR_OK;
    return generate_mips_from(ctx, p, first_level, stream);
}

tr_status tr_generate_mips_band(tr_context* ctx, const tr_pyramid* p, uint32_t y0, uint32_t y1, void* stream_) {
    if (!ctx || !p || !p->texels || p->levels < 3u || p->levels > TR_MAX_MIP_LEVELS) return TR_ERR_INVALID_ARGUMENT;
    // exact 2x2 boxes only: both sizes multiples of 4, the band on 4-row boundaries (its levels 1 and 2 then depend on
    // nothing but its own rows of level 0)
    if ((p->width & 3u) || (p->height & 3u) || (y0 & 3u) || y0 > y1 || y1 > p->height || ((y1 & 3u) && y1 != p->height))
        return TR_ERR_UNSUPPORTED;
    if (y0 == y1) return TR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    tr_mip_even_params ep;
    std::memset(&ep, 0, sizeof(ep));
    ep.w0 = p->width;
    ep.h0 = p->height;
    ep.nlevels = 2u;
    ep.src_offset = p->level_offset[0];
    ep.dst_offset[0] = p->level_offset[1];
    ep.dst_offset[1] = p->level_offset[2];
    ep.row_begin = y0 >> 1;
    ep.row_end = y1 >> 1;
    hipLaunchKernelGGL(mip_even_kernel, dim3(((p->width >> 1) + 15u) / 16u, ((ep.row_end - ep.row_begin) + 15u) / 16u), dim3(256), 0,
                       stream, (uint2*)p->texels, ep);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_set_tap_window(tr_context* ctx, uint32_t row_lo, uint32_t row_hi, void* excess_word_dev) {
    if (!ctx) return TR_ERR_INVALID_ARGUMENT;
    if (!excess_word_dev || row_hi == 0u) {   // off
        ctx->tap_row_lo = ctx->tap_row_hi = 0u;
        ctx->tap_excess = nullptr;
        return TR_OK;
    }
    if (row_lo >= row_hi || (row_lo & 1u) || ((uintptr_t)excess_word_dev & 3u)) return TR_ERR_INVALID_ARGUMENT;
    ctx->tap_row_lo = row_lo;
    ctx->tap_row_hi = row_hi;
    ctx->tap_excess = (uint32_t*)excess_word_dev;
    return TR_OK;
}

tr_status tr_shade_transmission(tr_context* ctx, const tr_gbuffer* g, const tr_uniforms* u,
                                const tr_push_constants* pc, const tr_pyramid* p, void* hdr_inout, tr_format format,
                                tr_rect rect, void* stream_) {
    if (!ctx || !g || !u || !pc || !p || !hdr_inout || !p->texels || !g->pos_depth || !g->nrm_scale ||
        !g->material_id || p->levels == 0 || p->levels > TR_MAX_MIP_LEVELS)
        return TR_ERR_INVALID_ARGUMENT;
    if (format != TR_FORMAT_RGBA16F && format != TR_FORMAT_RGBA32F) return TR_ERR_INVALID_ARGUMENT;
    if (!tables_ready(ctx, true)) return TR_ERR_TABLES_MISSING;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    tr_frame_params fp;
    tr_status st = ensure_slice_thresholds(ctx, u, stream);
    if (st != TR_OK) return st;
    st = fill_frame_params(ctx, g, u, pc, rect, &fp);
    if (st != TR_OK) return st;
    st = ensure_digested(ctx, stream);
    if (st != TR_OK) return st;
    st = ensure_levels(ctx, p, stream);
    if (st != TR_OK) return st;
    st = ensure_tap_records(ctx, fp.log2_fb_width, stream);
    if (st != TR_OK) return st;
    st = ensure_cluster_tables(ctx, u, fp.width, fp.height, stream);
    if (st != TR_OK) return st;
    st = check_textured_launch(ctx, g, fp);
    if (st != TR_OK) return st;
    st = tables_acquire(ctx, stream);
    if (st != TR_OK) return st;
    fp.pyr_levels = p->levels;
    // one-wave workgroups: blocks_per_xcd counts units of four waves
    tr_launch L;
    fill_launch(L, ctx, fp, g);
    dim3 grid(persistent_grid(ctx, fp.tiles_x * fp.tiles_y, L.vis != nullptr) * 4u), block(64);
    if (L.vis && ctx->front_list_hint) {   // (the frame recorder) walk the opaque launch's list of covered tiles
        L.front_list = ctx->d_front_list;
        L.front_list_count = ctx->d_front_list_count;
        L.front_list_cap = ctx->front_list_cap;
        // (a multiple of the sub-list count, at least one wave per sub-list: workgroup b walks list b % kFrontLists with stride
        //  grid / kFrontLists — a remainder would re-walk slot 0's tiles, fewer than kFrontLists waves would never advance)
        grid = dim3(std::max(kFrontLists, (ctx->num_cus * ctx->front_list_waves_per_cu + kFrontLists - 1u) / kFrontLists * kFrontLists));
    }
    {
        L.pyramid = (const uint2*)p->texels;
        L.hdr = hdr_inout;
        const bool half = format == TR_FORMAT_RGBA16F;
        if (ctx->any_textured) {
            launch_textured<true>(ctx, L, half, grid, block, stream);
#ifdef TR_TUNING_ENV
        } else if (ctx->lc_wgs_per_cu != 0u && !L.vis && !L.tile_cover) {
            // loader / consumer workgroups (tr_lc_kernels.h): G persistent workgroups per XCD
            L.fp.j_step = std::max(1u, ctx->num_cus / 8u * ctx->lc_wgs_per_cu);
            const dim3 lc_grid(8u * L.fp.j_step), lc_block(kLcWaves * 64u);
            if (half) hipLaunchKernelGGL(shade_lc_kernel<uint2>, lc_grid, lc_block, 0, stream, L);
            else hipLaunchKernelGGL(shade_lc_kernel<float4>, lc_grid, lc_block, 0, stream, L);
#endif
        } else {
            launch_shade<true, kTexNone>(L, half, grid, block, stream);
        }
    }
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

// ------------------------------------------------------------------------ the glam-pbr shading API (batched)
namespace {
// 2^24 elements per launch keeps every 32-bit byte offset of the 168-byte records below 4 GiB.
constexpr uint32_t kMaxBatch = 1u << 24;
inline bool batch_args_ok(const tr_context* ctx, uint32_t count, std::initializer_list<const void*> arrays) {
    if (!ctx || count > kMaxBatch) return false;
    for (const void* a : arrays)
        if (!a || ((uintptr_t)a & 3u)) return false;
    return true;
}
inline uint32_t batch_grid(uint32_t count) { return (count + 255u) / 256u; }
}  // namespace

tr_status tr_basic_brdf(tr_context* ctx, const void* params, uint32_t count, void* results, void* stream_) {
    if (!batch_args_ok(ctx, count, {params, results})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(basic_brdf_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_,
                       (const tr_basic_brdf_params*)params, count, (tr_brdf_result*)results);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_transmission_btdf(tr_context* ctx, const void* params, uint32_t count, void* rgb, void* stream_) {
    if (!batch_args_ok(ctx, count, {params, rgb})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(transmission_btdf_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_,
                       (const tr_transmission_btdf_params*)params, count, (float*)rgb);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_ibl_volume_refraction(tr_context* ctx, const void* params, uint32_t count, const tr_pyramid* p, void* rgb,
                                   void* stream_) {
    if (!batch_args_ok(ctx, count, {params, rgb}) || !p || !p->texels || p->levels == 0 || p->levels > TR_MAX_MIP_LEVELS)
        return TR_ERR_INVALID_ARGUMENT;
    if (!ctx->d_lut_pairs || ctx->lut_h == 0) return TR_ERR_TABLES_MISSING;   // the ggx_lut_sampler closure
    if (count == 0) return TR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    tr_status st = ensure_levels(ctx, p, stream);
    if (st != TR_OK) return st;
    st = tables_acquire(ctx, stream);
    if (st != TR_OK) return st;
    tr_ibl_tables t;
    t.pyramid = (const uint2*)p->texels;
    t.levels = ctx->d_levels;
    t.pyr_levels = p->levels;
    t.lut_pairs = ctx->d_lut_pairs;
    t.lut_width = ctx->lut_w;
    t.lut_height = ctx->lut_h;
    t.lut_stride = ctx->lut_stride;
    hipLaunchKernelGGL(ibl_volume_refraction_kernel, dim3(batch_grid(count)), dim3(256), 0, stream,
                       (const tr_ibl_volume_refraction_params*)params, count, t, (float*)rgb);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_ibl_volume_refraction_requests(tr_context* ctx, const void* params, uint32_t count, void* requests, void* stream_) {
    if (!batch_args_ok(ctx, count, {params, requests})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(ibl_requests_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_,
                       (const tr_ibl_volume_refraction_params*)params, count, (float*)requests);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_ibl_volume_refraction_resolve(tr_context* ctx, const void* params, uint32_t count, const void* framebuffer_rgb,
                                           const void* lut_ab, void* rgb, void* stream_) {
    if (!batch_args_ok(ctx, count, {params, framebuffer_rgb, lut_ab, rgb})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(ibl_resolve_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_,
                       (const tr_ibl_volume_refraction_params*)params, count, (const float*)framebuffer_rgb, (const float*)lut_ab,
                       (float*)rgb);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_light_direction_and_attenuation(tr_context* ctx, const void* fragment_position, const void* light_position,
                                             uint32_t count, void* out, void* stream_) {
    if (!batch_args_ok(ctx, count, {fragment_position, light_position, out})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(light_direction_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_,
                       (const float*)fragment_position, (const float*)light_position, count, (float*)out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_d_ggx(tr_context* ctx, const void* noh, const void* roughness, uint32_t count, void* out, void* stream_) {
    if (!batch_args_ok(ctx, count, {noh, roughness, out})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(d_ggx_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_, (const float*)noh,
                       (const float*)roughness, count, (float*)out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_v_smith_ggx_correlated(tr_context* ctx, const void* nov, const void* nol, const void* roughness, uint32_t count,
                                    void* out, void* stream_) {
    if (!batch_args_ok(ctx, count, {nov, nol, roughness, out})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(v_smith_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_, (const float*)nov,
                       (const float*)nol, (const float*)roughness, count, (float*)out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_fresnel_schlick(tr_context* ctx, const void* voh, const void* f0, const void* f90, uint32_t count, void* out,
                             void* stream_) {
    if (!batch_args_ok(ctx, count, {voh, f0, f90, out})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(fresnel_schlick_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_, (const float*)voh,
                       (const float*)f0, (const float*)f90, count, (float*)out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_compute_f0(tr_context* ctx, const void* metallic, const void* ior, const void* diffuse, uint32_t count,
                        void* out, void* stream_) {
    if (!batch_args_ok(ctx, count, {metallic, ior, diffuse, out})) return TR_ERR_INVALID_ARGUMENT;
    if (count == 0) return TR_OK;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(compute_f0_kernel, dim3(batch_grid(count)), dim3(256), 0, (hipStream_t)stream_, (const float*)metallic,
                       (const float*)ior, (const float*)diffuse, count, (float*)out);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

// ------------------------------------------------------------------------ multi-GPU: row bands + composite
}  // extern "C"

namespace {
struct rccl_api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;       // (optional: tr_comm_query)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
};
// Loaded once.  By soname first: a process that already mapped an RCCL (a framework's bundled copy has the same
// soname) gets that copy, so there is one RCCL per process.
const rccl_api& rccl() {
    static const rccl_api api = [] {
        rccl_api a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.handle) break;
        }
        if (!a.handle) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
        a.Broadcast = (decltype(a.Broadcast))dlsym(a.handle, "ncclBroadcast");
        a.Send = (decltype(a.Send))dlsym(a.handle, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(a.handle, "ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
        a.CommCount = (decltype(a.CommCount))dlsym(a.handle, "ncclCommCount");
        a.CommUserRank = (decltype(a.CommUserRank))dlsym(a.handle, "ncclCommUserRank");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.Broadcast && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
        return a;
    }();
    return api;
}
}  // namespace

struct tr_comm {
    ncclComm_t comm = nullptr;
    uint32_t nranks = 0, rank = 0;
    bool owned = false;
    int32_t last_error = 0;
};

extern "C" {

tr_status tr_band_rows(uint32_t height, uint32_t nranks, uint32_t rank, uint32_t* rows_per_rank, uint32_t* y0, uint32_t* y1) {
    if (!rows_per_rank || !y0 || !y1 || height == 0 || nranks == 0 || rank >= nranks) return TR_ERR_INVALID_ARGUMENT;
    const uint32_t even = (height + nranks - 1u) / nranks;
    const uint32_t rows = (even + kBlockTileH - 1u) / kBlockTileH * kBlockTileH;
    *rows_per_rank = rows;
    const uint64_t a = (uint64_t)rank * rows, b = a + rows;
    *y0 = (uint32_t)(a < height ? a : height);
    *y1 = (uint32_t)(b < height ? b : height);
    return TR_OK;
}

tr_status tr_comm_unique_id(uint8_t id_out[TR_COMM_ID_BYTES]) {
    static_assert(TR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return TR_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return TR_ERR_COMM;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return TR_ERR_COMM;
    std::memcpy(id_out, id.internal, TR_COMM_ID_BYTES);
    return TR_OK;
}

tr_status tr_comm_create(tr_context* ctx, const uint8_t id_in[TR_COMM_ID_BYTES], uint32_t nranks, uint32_t rank, tr_comm** out) {
    if (!ctx || !id_in || !out || nranks == 0 || rank >= nranks) return TR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!rccl().ok) return TR_ERR_COMM;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    tr_comm* c = new (std::nothrow) tr_comm();
    if (!c) return TR_ERR_OUT_OF_MEMORY;
    ncclUniqueId id;
    std::memcpy(id.internal, id_in, TR_COMM_ID_BYTES);
    const ncclResult_t r = rccl().CommInitRank(&c->comm, (int)nranks, id, (int)rank);
    if (r != ncclSuccess) {
        ctx->last_hip_error = (int32_t)r;   // (no communicator to hold it)
        delete c;
        return TR_ERR_COMM;
    }
    c->nranks = nranks;
    c->rank = rank;
    c->owned = true;
    *out = c;
    return TR_OK;
}

tr_status tr_comm_from_nccl(void* nccl_comm, uint32_t nranks, uint32_t rank, tr_comm** out) {
    if (!nccl_comm || !out || nranks == 0 || rank >= nranks) return TR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!rccl().ok) return TR_ERR_COMM;
    tr_comm* c = new (std::nothrow) tr_comm();
    if (!c) return TR_ERR_OUT_OF_MEMORY;
    c->comm = (ncclComm_t)nccl_comm;
    c->nranks = nranks;
    c->rank = rank;
    c->owned = false;
    *out = c;
    return TR_OK;
}

tr_status tr_comm_destroy(tr_comm* comm) {
    if (!comm) return TR_ERR_INVALID_ARGUMENT;
    tr_status st = TR_OK;
    if (comm->owned && comm->comm && rccl().ok && rccl().CommDestroy(comm->comm) != ncclSuccess) st = TR_ERR_COMM;
    delete comm;
    return st;
}

int32_t tr_comm_last_error(const tr_comm* comm) { return comm ? comm->last_error : (rccl().ok ? 0 : -1); }

tr_status tr_comm_query(tr_comm* comm, uint32_t* nranks_out, uint32_t* rank_out) {
    if (!comm || !nranks_out || !rank_out) return TR_ERR_INVALID_ARGUMENT;
    if (!rccl().ok || !rccl().CommCount || !rccl().CommUserRank || !comm->comm) return TR_ERR_COMM;
    int n = 0, r = 0;
    ncclResult_t e = rccl().CommCount(comm->comm, &n);
    if (e == ncclSuccess) e = rccl().CommUserRank(comm->comm, &r);
    if (e != ncclSuccess) {
        comm->last_error = (int32_t)e;
        return TR_ERR_COMM;
    }
    *nranks_out = (uint32_t)n;
    *rank_out = (uint32_t)r;
    return TR_OK;
}

tr_status tr_allgather_frame(tr_context* ctx, tr_comm* comm, void* frame, uint32_t width, uint32_t rows_per_rank,
                             tr_format format, void* stream_) {
    if (!ctx || !comm || !frame || width == 0 || rows_per_rank == 0) return TR_ERR_INVALID_ARGUMENT;
    if (format != TR_FORMAT_RGBA16F && format != TR_FORMAT_RGBA32F && format != TR_FORMAT_RGBA8 && format != TR_FORMAT_RGB8) return TR_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return TR_ERR_COMM;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t band_bytes = (size_t)width * rows_per_rank * (format == TR_FORMAT_RGBA16F ? 8u : format == TR_FORMAT_RGBA32F ? 16u : format == TR_FORMAT_RGB8 ? 3u : 4u);
    char* base = static_cast<char*>(frame);
    // in place: this rank's band already sits at its slot of the receive buffer
    const ncclResult_t r = rccl().AllGather(base + (size_t)comm->rank * band_bytes, base, band_bytes, ncclUint8, comm->comm,
                                            (hipStream_t)stream_);
    if (r != ncclSuccess) {
        comm->last_error = (int32_t)r;
        return TR_ERR_COMM;
    }
    return TR_OK;
}

tr_status tr_halo_rows(uint32_t total_rows, uint32_t rows_per_rank, uint32_t nranks, uint32_t owner, uint32_t reader, uint32_t halo_rows,
                       uint32_t* y0, uint32_t* y1) {
    if (!y0 || !y1 || total_rows == 0 || rows_per_rank == 0 || nranks == 0 || owner >= nranks || reader >= nranks)
        return TR_ERR_INVALID_ARGUMENT;
    auto band = [&](uint32_t r, uint64_t& a, uint64_t& b) {
        a = std::min<uint64_t>((uint64_t)r * rows_per_rank, total_rows);
        b = std::min<uint64_t>((uint64_t)(r + 1u) * rows_per_rank, total_rows);
    };
    uint64_t oa, ob, ra, rb;
    band(owner, oa, ob);
    band(reader, ra, rb);
    const uint64_t wa = ra > halo_rows ? ra - halo_rows : 0u, wb = std::min<uint64_t>(rb + halo_rows, total_rows);   // the reader's window
    const uint64_t a = std::max(oa, wa), b = std::min(ob, wb);
    *y0 = (uint32_t)(a < b ? a : 0u);
    *y1 = (uint32_t)(a < b ? b : 0u);
    if (ra == rb) *y0 = *y1 = 0u;   // (a reader without rows reads nothing)
    return TR_OK;
}

tr_status tr_exchange_halo(tr_context* ctx, tr_comm* comm, void* level_rows, uint32_t row_bytes, uint32_t total_rows,
                           uint32_t rows_per_rank, uint32_t halo_rows, void* stream_) {
    if (!ctx || !comm || !level_rows || row_bytes == 0 || total_rows == 0 || rows_per_rank == 0) return TR_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return TR_ERR_COMM;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    char* base = static_cast<char*>(level_rows);
    // every rank sends the rows of its band that lie in a peer's window and receives the rows of the peer's band that lie in
    // its own, in place, as one group: the links carry band-border rows only (halo_rows >= the frame: an all-gather)
    ncclResult_t r = rccl().GroupStart();
    for (uint32_t peer = 0; r == ncclSuccess && peer < comm->nranks; ++peer) {
        if (peer == comm->rank) continue;
        uint32_t a = 0, b = 0;
        tr_halo_rows(total_rows, rows_per_rank, comm->nranks, comm->rank, peer, halo_rows, &a, &b);    // mine, for the peer
        if (b > a) r = rccl().Send(base + (size_t)a * row_bytes, (size_t)(b - a) * row_bytes, ncclUint8, (int)peer, comm->comm, (hipStream_t)stream_);
        if (r != ncclSuccess) break;
        tr_halo_rows(total_rows, rows_per_rank, comm->nranks, peer, comm->rank, halo_rows, &a, &b);    // the peer's, for me
        if (b > a) r = rccl().Recv(base + (size_t)a * row_bytes, (size_t)(b - a) * row_bytes, ncclUint8, (int)peer, comm->comm, (hipStream_t)stream_);
    }
    const ncclResult_t e = rccl().GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) {
        comm->last_error = (int32_t)r;
        return TR_ERR_COMM;
    }
    return TR_OK;
}

tr_status tr_set_strips(tr_context* ctx, uint32_t strip_rows, uint32_t nranks, uint32_t rank) {
    if (!ctx) return TR_ERR_INVALID_ARGUMENT;
    if (strip_rows == 0u || nranks <= 1u) {   // off
        ctx->strip_rows = 0;
        ctx->strip_world = 1;
        ctx->strip_rank = 0;
        return TR_OK;
    }
    if (strip_rows % kBlockTileH != 0u || rank >= nranks) return TR_ERR_INVALID_ARGUMENT;
    ctx->strip_rows = strip_rows;
    ctx->strip_world = nranks;
    ctx->strip_rank = rank;
    return TR_OK;
}

tr_status tr_strip_of_rank(uint32_t height, uint32_t strip_rows, uint32_t nranks, uint32_t rank, uint32_t k, uint32_t* y0,
                           uint32_t* y1) {
    if (!y0 || !y1 || height == 0 || strip_rows == 0 || nranks == 0 || rank >= nranks) return TR_ERR_INVALID_ARGUMENT;
    const uint64_t a = ((uint64_t)k * nranks + rank) * strip_rows, b = a + strip_rows;
    *y0 = (uint32_t)(a < height ? a : height);
    *y1 = (uint32_t)(b < height ? b : height);
    return TR_OK;
}

tr_status tr_allgather_strips(tr_context* ctx, tr_comm* comm, void* frame, uint32_t width, uint32_t height, uint32_t strip_rows,
                              tr_format format, void* stream_) {
    if (!ctx || !comm || !frame || width == 0 || height == 0 || strip_rows == 0) return TR_ERR_INVALID_ARGUMENT;
    if (format != TR_FORMAT_RGBA16F && format != TR_FORMAT_RGBA32F && format != TR_FORMAT_RGBA8 && format != TR_FORMAT_RGB8) return TR_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return TR_ERR_COMM;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t row_bytes = (size_t)width * (format == TR_FORMAT_RGBA16F ? 8u : format == TR_FORMAT_RGBA32F ? 16u : format == TR_FORMAT_RGB8 ? 3u : 4u);
    char* base = static_cast<char*>(frame);
    // every strip is broadcast in place from the rank that shaded it; one group: RCCL schedules the strips of all roots
    // together (each xGMI link carries the strips of one peer in each direction, like the band all-gather)
    ncclResult_t r = rccl().GroupStart();
    for (uint32_t s = 0, y = 0; r == ncclSuccess && y < height; ++s, y += strip_rows) {
        const size_t bytes = (size_t)((height - y < strip_rows) ? height - y : strip_rows) * row_bytes;
        char* at = base + (size_t)y * row_bytes;
        r = rccl().Broadcast(at, at, bytes, ncclUint8, (int)(s % comm->nranks), comm->comm, (hipStream_t)stream_);
    }
    const ncclResult_t e = rccl().GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) {
        comm->last_error = (int32_t)r;
        return TR_ERR_COMM;
    }
    return TR_OK;
}


tr_status tr_lottes_defaults(tr_lottes_params* out) {
    if (!out) return TR_ERR_INVALID_ARGUMENT;
    // T. Lottes, "Advanced Techniques and Optimization of HDR Color Pipelines" (GDC 2016) reference values
    out->contrast = 1.6f;
    out->shoulder = 0.977f;
    out->hdr_max = 8.0f;
    out->mid_in = 0.18f;
    out->mid_out = 0.267f;
    out->crosstalk = 4.0f;
    out->saturation = out->contrast;
    out->cross_saturation = out->contrast * 16.0f;
    return TR_OK;
}

tr_status tr_bake_lottes_params(const tr_lottes_params* q, tr_tonemap_params* out) {
    if (!q || !out) return TR_ERR_INVALID_ARGUMENT;
    const float a = q->contrast, d = q->shoulder;
    const float mid_in_a = std::pow(q->mid_in, a), mid_in_ad = std::pow(q->mid_in, a * d);
    const float max_a = std::pow(q->hdr_max, a), max_ad = std::pow(q->hdr_max, a * d);
    const float denom = (max_ad - mid_in_ad) * q->mid_out;
    out->a = a;
    out->d = d;
    out->b = (-mid_in_a + max_a * q->mid_out) / denom;
    out->c = (max_ad * mid_in_a - max_a * mid_in_ad * q->mid_out) / denom;
    out->crosstalk = q->crosstalk;
    out->saturation = q->saturation;
    out->cross_saturation = q->cross_saturation;
    return TR_OK;
}

tr_status tr_tonemap(tr_context* ctx, const void* hdr, uint32_t width, uint32_t height, const tr_tonemap_params* params,
                     void* out_rgba8, int32_t bgra, void* stream_) {
    if (!ctx || !hdr || !params || !out_rgba8 || width == 0 || height == 0) return TR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = width * height;
    if (((uintptr_t)hdr & 15u) || ((uintptr_t)out_rgba8 & 7u)) return TR_ERR_INVALID_ARGUMENT;   // 2 pixels per thread
    hipLaunchKernelGGL(tonemap_kernel, dim3((n + 511u) / 512u), dim3(256), 0, stream, (const uint2*)hdr,
                       (uint32_t*)out_rgba8, n, *params, (int)bgra, params->saturation / params->cross_saturation);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

tr_status tr_tonemap_rgb8(tr_context* ctx, const void* hdr, uint32_t width, uint32_t height, const tr_tonemap_params* params,
                          void* out_rgb8, int32_t bgra, void* stream_) {
    if (!ctx || !hdr || !params || !out_rgb8 || width == 0 || height == 0) return TR_ERR_INVALID_ARGUMENT;
    const uint32_t n = width * height;
    if ((n & 3u) || ((uintptr_t)hdr & 15u) || ((uintptr_t)out_rgb8 & 3u)) return TR_ERR_INVALID_ARGUMENT;   // 4 pixels per thread
    hipStream_t stream = (hipStream_t)stream_;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(tonemap_rgb8_kernel, dim3((n + 1023u) / 1024u), dim3(256), 0, stream, (const uint2*)hdr, (uint32_t*)out_rgb8, n,
                       *params, (int)bgra, params->saturation / params->cross_saturation);
    TR_HIP(ctx, hipGetLastError());
    return TR_OK;
}

}  // extern "C"

namespace {
// Optional GPU timestamps around the passes of a frame (tr_record_frame_timed): one hipEvent pair per zone.
struct zone_recorder {
    struct zone { const char* name; hipEvent_t begin, end; };
    std::vector<zone> zones;
    hipStream_t stream;
    bool failed = false;
    int open(const char* name) {
        zone z{name, nullptr, nullptr};
        if (hipEventCreate(&z.begin) != hipSuccess || hipEventCreate(&z.end) != hipSuccess ||
            hipEventRecord(z.begin, stream) != hipSuccess)
            failed = true;
        zones.push_back(z);
        return (int)zones.size() - 1;
    }
    void close(int i) {
        if (zones[(size_t)i].end && hipEventRecord(zones[(size_t)i].end, stream) != hipSuccess) failed = true;
    }
    ~zone_recorder() {
        for (auto& z : zones) {
            if (z.begin) (void)hipEventDestroy(z.begin);
            if (z.end) (void)hipEventDestroy(z.end);
        }
    }
};
struct zone_scope {   // no-op without a recorder
    zone_recorder* r;
    int i = -1;
    zone_scope(zone_recorder* rec, const char* name) : r(rec) { if (r) i = r->open(name); }
    void close() { if (r && i >= 0) { r->close(i); i = -1; } }
    ~zone_scope() { close(); }
};

tr_status record_frame(tr_context* ctx, const tr_frame_desc* f, void* stream, zone_recorder* rec) {
    if (!ctx || !f || !f->push || !f->uniforms || !f->culling || !f->view_matrix || !f->view_rotation ||
        !f->cluster_aabbs || !f->cluster_light_counts || !f->light_indices || !f->hdr || !f->pyramid.texels ||
        f->num_clusters == 0 || ((f->tonemap == nullptr) != (f->ldr_out == nullptr)))
        return TR_ERR_INVALID_ARGUMENT;
    if (f->ldr_out && f->hdr_format != TR_FORMAT_RGBA16F) return TR_ERR_INVALID_ARGUMENT;   // the tonemap reads RGBA16F
    const uint32_t w = f->push->framebuffer_size[0], h = f->push->framebuffer_size[1];
    if (f->pyramid.width != w || f->pyramid.height != h) return TR_ERR_INVALID_ARGUMENT;
    if (w == 0 || h == 0 || w > 65535u || h > 65535u) return TR_ERR_INVALID_ARGUMENT;
    if (!ctx->d_position) return TR_ERR_TABLES_MISSING;
    if (ctx->strip_rows != 0u) return TR_ERR_UNSUPPORTED;   // (the frame recorder renders whole frames)
    if (rec && stream_is_capturing(stream)) return TR_ERR_UNSUPPORTED;   // (the timed frame synchronises around every pass)
    ctx->cover_cleared = false;
    // RGBA16F frames are shaded straight from the rasteriser's visibility words (shade_kernel's VIS launches): no resolve,
    // the work planes of the descriptor stay untouched.  RGBA32F frames go through the planes.
    const bool use_vis = f->hdr_format == TR_FORMAT_RGBA16F;
    zone_scope all(rec, "all commands");
    tr_status st;
    void* draws[TR_NUM_DRAW_BUFFERS] = {ctx->d_draws[0], ctx->d_draws[1], ctx->d_draws[2], ctx->d_draws[3]};
    if (!rec) {
        // The untimed frame fuses its launch-latency-bound front end into ONE launch (frame_front_kernel): culling, light
        // assignment and the coverage clear side by side, and behind the LAST culling workgroup (a ticket counted with
        // agent-scope atomics; what it reads are the atomics' own words) the demultiplex and both layers' draw scans; the
        // instance counts are zeroed by their last reader instead of a fill.  tests/test_gpu_culling.py holds the hand-over
        // against the unfused passes over many frames.  (The timed frame launches every pass on its own, below.)
        hipStream_t s_ = (hipStream_t)stream;
        TR_HIP(ctx, hipSetDevice(ctx->device));
        // (a captured frame always zeroes its counts: a replay may follow a direct tr_frustum_culling that left them set)
        if (!ctx->counts_clean || stream_is_capturing(s_)) TR_HIP(ctx, zero_fill(ctx->d_instance_counts, sizeof(uint32_t) * ctx->num_primitives, s_));
        ctx->counts_clean = false;
        tr_cull_params cp;
        cp.pc = *f->culling;
        cp.num_instances = ctx->num_instances;
        cp.num_primitives = ctx->num_primitives;
        tr_assign_params ap;
        std::memcpy(ap.view_matrix, f->view_matrix, sizeof(ap.view_matrix));
        std::memcpy(ap.view_rotation, f->view_rotation, sizeof(ap.view_rotation));
        ap.num_lights = ctx->num_lights;
        ap.num_clusters = f->num_clusters;
        const uint32_t cull_blocks = (ctx->num_instances + 255u) / 256u, assign_blocks = (f->num_clusters + 3u) / 4u;
        st = ensure_vis_buffers(ctx, w, h, stream);
        if (st != TR_OK) return st;
        const uint32_t clear_vectors = (uint32_t)(cover_clear_bytes(w, h) / 16u), clear_blocks = (clear_vectors + 255u) / 256u;
        tr_front_demux dm;
        dm.ticket = ctx->d_front_ticket;
        dm.primitives = ctx->d_primitives;
        dm.num_primitives = ctx->num_primitives;
        dm.draw_counts = ctx->d_draw_counts;
        for (uint32_t k = 0; k < TR_NUM_DRAW_BUFFERS; ++k) dm.out.draws[k] = ctx->d_draws[k];
        {
            const tr_gbuffer_target* const layer_targets[2] = {&f->opaque_layer, &f->transmissive_layer};
            fill_two_layers(ctx, draws, layer_targets, dm.two);
        }
        hipLaunchKernelGGL(frame_front_kernel, dim3(cull_blocks + assign_blocks + clear_blocks), dim3(256), 0, s_, cp,
                           (const tr_primitive_info*)ctx->d_primitives, (const tr_instance*)ctx->d_instances,
                           ctx->d_instance_counts, cull_blocks, ap, (const tr_alight*)ctx->d_alights,
                           (const tr_cluster_aabb*)f->cluster_aabbs, (uint32_t*)f->cluster_light_counts, (uint32_t*)f->light_indices,
                           assign_blocks, (uint4*)ctx->d_tile_cover[0], clear_vectors, dm);
        TR_HIP(ctx, hipGetLastError());
        ctx->cover_cleared = true;
        st = tr_set_cluster_tables(ctx, f->cluster_light_counts, f->light_indices, f->num_clusters);
        if (st != TR_OK) return st;
        st = rasterize_impl(ctx, ctx->d_draw_counts, draws, f->push, &f->opaque_layer, &f->transmissive_layer, stream, true, !use_vis);
        if (st != TR_OK) return st;
        ctx->counts_clean = true;   // (the demultiplex behind the culling blocks zeroed what it read)
    } else {
    {   // "frustum culling" (zeroing the counts + "frustum culling compute shader")
        zone_scope z(rec, "frustum culling");
        st = tr_frustum_culling(ctx, ctx->d_primitives, ctx->num_primitives, ctx->d_instances, ctx->num_instances,
                                f->culling, ctx->d_instance_counts, stream);
    }
    if (st != TR_OK) return st;
    {
        zone_scope z(rec, "assign lights to clusters");
        st = tr_assign_lights_to_clusters(ctx, f->view_matrix, f->view_rotation, f->cluster_aabbs, f->num_clusters,
                                          f->cluster_light_counts, f->light_indices, stream);
    }
    if (st != TR_OK) return st;
    st = tr_set_cluster_tables(ctx, f->cluster_light_counts, f->light_indices, f->num_clusters);
    if (st != TR_OK) return st;
    {
        zone_scope z(rec, "demultiplex draws compute shader");
        st = tr_demultiplex_draws(ctx, ctx->d_primitives, ctx->num_primitives, ctx->d_instance_counts, ctx->d_draw_counts,
                                  draws, stream);
    }
    if (st != TR_OK) return st;
    {   // the visibility-buffer rasteriser: stands for the depth pre-passes and the EQUAL-tested colour-pass draws
        zone_scope z(rec, "depth pre pass");
        st = rasterize_impl(ctx, ctx->d_draw_counts, draws, f->push, &f->opaque_layer, &f->transmissive_layer, stream, false, !use_vis);
    }
    if (st != TR_OK) return st;
    }
    // "main opaque" -> "opaque framebuffer mipchain" -> "opaque transmissive objects"
    tr_gbuffer layers[2];
    const tr_gbuffer_target* targets[2] = {&f->opaque_layer, &f->transmissive_layer};
    for (int k = 0; k < 2; ++k) {
        layers[k].pos_depth = targets[k]->pos_depth;
        layers[k].nrm_scale = targets[k]->nrm_scale;
        layers[k].uv = targets[k]->uv;
        layers[k].material_id = targets[k]->material_id;
        layers[k].width = w;
        layers[k].height = h;
        layers[k].origin_x = layers[k].origin_y = 0;
    }
    const tr_rect whole = {0u, 0u, w, h};
    // Level 1 of the opaque pyramid comes out of the opaque launches themselves when they shade from visibility words and
    // both frame sizes are even (then the blit is the 2x2 box of a wave tile's own quads): the mip chain starts at level 2.
    const bool fused_level1 = use_vis && f->pyramid.levels >= 2u && (w & 1u) == 0u && (h & 1u) == 0u;
    // A frame that is presented and not timed pass by pass: its VIS launches tonemap the pixels whose final colour they write
    // (shade_kernel, `present`) — there is no tonemap pass.  The timed recorder keeps the reference's separate pass (its zone).
    const bool present_in_passes = use_vis && f->ldr_out != nullptr && rec == nullptr;
    if (f->ldr_out && (((uintptr_t)f->hdr & 15u) || ((uintptr_t)f->ldr_out & 7u))) return TR_ERR_INVALID_ARGUMENT;
    if (present_in_passes) {
        ctx->present_hint = (uint32_t*)f->ldr_out;
        ctx->present_params_hint = *f->tonemap;
        ctx->present_bgra_hint = (int32_t)f->bgra;
    }
    ctx->front_list_hint = use_vis && ctx->front_list_waves_per_cu != 0u;
    struct present_guard {   // (the hints never outlive the call)
        tr_context* c;
        ~present_guard() { c->present_hint = nullptr; c->front_list_hint = false; }
    } present_scope{ctx};
    {
        zone_scope z(rec, "main opaque");
        ctx->cover_hint = ctx->d_tile_cover[0];
        ctx->vis_hint = use_vis ? ctx->d_vis[0] : nullptr;
        ctx->planes_hint = ctx->d_tri_planes;
        ctx->vis_front_hint = use_vis ? ctx->d_vis[1] : nullptr;
        ctx->cover_front_hint = ctx->d_tile_cover[1];
        ctx->mip1_hint = fused_level1 ? (void*)((uint2*)f->pyramid.texels + f->pyramid.level_offset[1]) : nullptr;
        st = tr_shade_opaque(ctx, &layers[0], f->uniforms, f->push, f->hdr, f->hdr_format, f->pyramid.texels, whole, stream);
        ctx->mip1_hint = nullptr;
        ctx->cover_hint = nullptr;
        ctx->vis_hint = nullptr;
        ctx->planes_hint = nullptr;
        ctx->vis_front_hint = nullptr;
        ctx->cover_front_hint = nullptr;
    }
    if (st != TR_OK) return st;
    {
        zone_scope z(rec, "opaque framebuffer mipchain");
        st = generate_mips_from(ctx, &f->pyramid, fused_level1 ? 2u : 1u, stream);
    }
    if (st != TR_OK) return st;
    {
        zone_scope z(rec, "opaque transmissive objects");
        ctx->cover_hint = ctx->d_tile_cover[1];
        ctx->vis_hint = use_vis ? ctx->d_vis[1] : nullptr;
        ctx->planes_hint = ctx->d_tri_planes + ctx->work_capacity;
        st = tr_shade_transmission(ctx, &layers[1], f->uniforms, f->push, &f->pyramid, f->hdr, f->hdr_format, whole, stream);
        ctx->cover_hint = nullptr;
        ctx->vis_hint = nullptr;
        ctx->planes_hint = nullptr;
    }
    if (st != TR_OK) return st;
    if (use_vis) ctx->vis_clean = true;   // (both passes enqueued: every visibility word the frame set is zeroed again)
    if (f->ldr_out && !present_in_passes) {
        zone_scope z(rec, "tonemapping");
        // (the tiles no fragment of either layer landed in hold the clear colour: tonemapped once per workgroup)
        tr_tonemap_tiles tt;
        tt.cover[0] = ctx->d_tile_cover[0];
        tt.cover[1] = ctx->d_tile_cover[1];
        tt.width = w;
        tt.height = h;
        tt.tiles_x = (w + 63u) / 64u;
        hipLaunchKernelGGL(tonemap_tiles_kernel, dim3((tt.tiles_x + 1u) / 2u, (h + 3u) / 4u), dim3(256), 0, (hipStream_t)stream,
                           (const uint2*)f->hdr, (uint32_t*)f->ldr_out, *f->tonemap, (int)f->bgra, tt,
                           f->tonemap->saturation / f->tonemap->cross_saturation);
        TR_HIP(ctx, hipGetLastError());
        st = TR_OK;
    }
    return st;
}
}  // namespace

extern "C" {

tr_status tr_record_frame(tr_context* ctx, const tr_frame_desc* f, void* stream) { return record_frame(ctx, f, stream, nullptr); }

tr_status tr_record_frame_timed(tr_context* ctx, const tr_frame_desc* f, void* stream, tr_frame_zone* zones_out, uint32_t capacity,
                                uint32_t* num_zones_out) {
    if (!ctx || !zones_out || !num_zones_out) return TR_ERR_INVALID_ARGUMENT;
    *num_zones_out = 0;
    TR_HIP(ctx, hipSetDevice(ctx->device));
    zone_recorder rec;
    rec.stream = (hipStream_t)stream;
    const tr_status st = record_frame(ctx, f, stream, &rec);
    if (st != TR_OK) return st;
    if (rec.failed) return TR_ERR_HIP;
    TR_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    uint32_t n = 0;
    for (const auto& z : rec.zones) {
        if (n >= capacity) break;
        float ms = 0.0f;
        TR_HIP(ctx, hipEventElapsedTime(&ms, z.begin, z.end));
        zones_out[n].name = z.name;
        zones_out[n].milliseconds = ms;
        zones_out[n]._pad = 0;
        ++n;
    }
    *num_zones_out = n;
    return TR_OK;
}

}  // extern "C"

#ifdef TR_RASTER_TIMING
#define TR_RASTER_PROBE_HOST 1
#include "tr_raster_probe.h"   // (profiling builds only) tr_debug_read_raster_timing
#endif
