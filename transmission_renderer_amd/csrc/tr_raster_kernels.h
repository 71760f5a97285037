// tr_raster_kernels.h — geometry front end for gfx950: vertex stage, triangle setup, a
// visibility-buffer rasteriser and the resolve into the two TGB-v1 layers the shading kernels read.
//
// Reference behaviour being replaced (file:line relative to the reference root):
//   vertex_instanced[_with_scale], depth_pre_pass_*        shader/src/lib.rs:269-385
//   depth pre-pass (GREATER, write) + colour pass (EQUAL)  src/pipelines.rs:309-398, src/main.rs:1900-2042
//   indirect draws from the demultiplexed buffers          src/main.rs:1811-1838, 1921-2040
// The fixed-function part (clipping, rasterisation rules, interpolation) has no reference source; it is restated
// here exactly as in oracle/tr_oracle.c `o_rasterize` (same fp32 operations, contraction off), so coverage,
// depth and every interpolated attribute are bit-identical to the CPU restatement.
//
// Shape of the work (one layer = two of the four draw buffers):
//   scan_draws     one workgroup: triangles per draw -> prefix, so triangle t of the layer's draw stream is known
//   setup          one thread per triangle: vertex stage, clip-space edge functions, bounds, work-item count
//   scan_items     work items per triangle -> prefix (chunk sums, prefix of the sums, per-chunk scan)
//   raster         persistent waves; a work item is one 8-pixel-tall row of a triangle's bounds, at most 32 pixels
//                  wide; a wave covers it in 8x8 pixel blocks and resolves visibility with one 64-bit atomicMax of
//                  (depth bits << 32 | t) per covered pixel: reversed-Z GREATER, and among equal depths the
//                  later-drawn triangle wins, independent of execution order (deterministic)
//   resolve        one thread per pixel: winner -> barycentrics -> TGB-v1 planes
#pragma once

#include "tr_geometry_kernels.h"
#include "tr_visibility.h"
#include "tr_texture_kernels.h"

namespace tr {

struct tr_geometry_view {
    const float* position;   // 3 floats per vertex
    const float* normal;
    const float* uv;         // 2 floats per vertex
    const uint32_t* index;
    const tr_instance* instances;
};

struct tr_layer_counts {      // written by scan_draws / scan_items, read by the later kernels of the layer
    uint32_t num_draws_first;  // draws taken from the layer's first buffer
    uint32_t num_draws;        // ... from both
    uint32_t num_triangles;
    uint32_t num_items;
};

// (a work item spans at most 4 8x8 blocks horizontally: measured on the 4K mesh / glTF demo frames — 16: 207 / 205 us,
//  8: 197 / 197, 4: 193 / 195, 3: 191 / 196, 2: 197 / 200, 1: 209 / 213; short items balance and hide each other's latency)
constexpr uint32_t kItemWidthBlocks = 4u;
constexpr uint32_t kItemTileColumns = (kItemWidthBlocks + 7u) / 8u + 1u;   // 64-pixel tile columns an item can span

// ------------------------------------------------------------------------ block-wide exclusive scan
// NT threads (1024, or 256 inside the frame's first launch); returns the exclusive prefix of `v` over the block and the
// block total (in every thread).
template <uint32_t NT = 1024u>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave /*[17]*/, uint32_t& total) {
    constexpr int kWaves = (int)(NT / 64u);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    if (lane == 63u) lds_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t t = lds_wave[w];
            lds_wave[w] = run;
            run += t;
        }
        lds_wave[16] = run;
    }
    __syncthreads();
    const uint32_t result = lds_wave[wave] + incl - v;
    total = lds_wave[16];
    __syncthreads();
    return result;
}

// tri_base[d] = number of triangles before draw d of the layer's stream (first buffer's draws, then the second's)
template <uint32_t NT = 1024u>
__device__ __forceinline__ void raster_scan_draws_body(const tr_draw_command* __restrict__ draws_a,
                                                                 const tr_draw_command* __restrict__ draws_b,
                                                                 const uint32_t* __restrict__ draw_counts, uint32_t buffer_a,
                                                                 uint32_t capacity_draws, uint32_t capacity_triangles,
                                                                 uint32_t* __restrict__ tri_base,
                                                                 tr_layer_counts* __restrict__ counts) {
    __shared__ uint32_t lds[17];
    const uint32_t na = min(draw_counts[buffer_a], capacity_draws);
    const uint32_t nb = min(draw_counts[buffer_a + 1u], capacity_draws - na);
    const uint32_t n = na + nb;
    uint32_t running = 0;
    for (uint32_t base = 0; base < n; base += NT) {
        const uint32_t d = base + threadIdx.x;
        uint32_t tris = 0;
        if (d < n) {
            const tr_draw_command c = d < na ? draws_a[d] : draws_b[d - na];
            const uint64_t t64 = (uint64_t)(c.index_count / 3u) * c.instance_count;
            tris = (uint32_t)min(t64, (uint64_t)capacity_triangles);
        }
        uint32_t total;
        const uint32_t ex = block_exclusive_scan<NT>(tris, lds, total);
        if (d < n) tri_base[d] = min(running + ex, capacity_triangles);
        running = min(running + total, capacity_triangles);
    }
    if (threadIdx.x == 0) {
        tri_base[n] = running;
        counts->num_draws_first = na;
        counts->num_draws = n;
        counts->num_triangles = running;   // never above the capacity the host sized the records for
        counts->num_items = 0;
    }
}

// largest i in [0, n) with base[i] <= x  (base ascending, base[0] == 0)
__device__ __forceinline__ uint32_t upper_index(const uint32_t* __restrict__ base, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (base[mid] <= x) lo = mid;
        else hi = mid;
    }
    return lo;
}

// The same index by an 8-ary search: seven pivots are loaded together per step, so 1 000 entries take 4 dependent round
// trips instead of 10 and 200 000 take 6 instead of 18.  `base` has n + 1 entries and base[n] > x (the total): a pivot
// beyond the interval is clamped to its end, whose entry is above x by the search's invariant — no bounds test per pivot.
__device__ __forceinline__ uint32_t upper_index_wide(const uint32_t* __restrict__ base, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        const uint32_t step = (hi - lo + 7u) >> 3;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t j = 1; j <= 7u; ++j) c += ld<uint32_t>(base, min(lo + j * step, hi) * 4u) <= x ? 1u : 0u;
        lo += c * step;
        hi = min(lo + step, hi);
    }
    return lo;
}

struct tr_raster_frame {
    float proj_view[16];
    uint32_t width, height;
};

// vertex_instanced_with_scale for one vertex: world position, clip position (and the rotated normal when asked)
__device__ __forceinline__ void vertex_stage(const tr_geometry_view& g, const tr_instance& inst, const float* pv,
                                             uint32_t vi, float world[3], float clip[4]) {
#pragma clang fp contract(off)
    similarity_apply(inst, g.position[vi * 3u], g.position[vi * 3u + 1u], g.position[vi * 3u + 2u], world);
    mat4_mul_point(pv, world[0], world[1], world[2], clip);
}

// One triangle of the layer's draw stream: record + planes written, returns its number of work items.
__device__ __forceinline__ uint32_t raster_setup_body(const tr_geometry_view g, const tr_raster_frame f,
                                                           const tr_draw_command* __restrict__ draws_a,
                                                           const tr_draw_command* __restrict__ draws_b,
                                                           const uint32_t* __restrict__ tri_base,
                                                           const tr_layer_counts* __restrict__ counts, uint32_t alpha_buffer_b,
                                                           tr_tri_record* __restrict__ records,
                                                           tr_tri_planes* __restrict__ tri_planes,
                                                           const uint32_t* __restrict__ material_flags /* tr_dmat::flags or null */,
                                                           uint32_t flags_stride /* in words */, uint32_t t) {
#pragma clang fp contract(off)
    // (8-ary: the draw of a triangle is the first of the set-up's eight or so dependent round trips, and the launch is nothing but
    //  that chain — 960 triangles at 4K; tri_base[num_draws] is the total, above every t)
    const uint32_t d = upper_index_wide(tri_base, counts->num_draws, t);
    const bool second = d >= counts->num_draws_first;
    const tr_draw_command c = second ? draws_b[d - counts->num_draws_first] : draws_a[d];
    const uint32_t local = t - tri_base[d], ntri = c.index_count / 3u;
    const uint32_t inst_id = c.first_instance + local / ntri, tri = local % ntri;
    const tr_instance inst = g.instances[inst_id];
    tr_tri_record r;
    float X[3], Y[3], W[3], P[3][3], N[3][3];   // (P, N: the vertex stage's world position and rotated normal per corner)
    const float hw = 0.5f * (float)f.width, hh = 0.5f * (float)f.height;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t vi = g.index[c.first_index + tri * 3u + (uint32_t)k] + (uint32_t)c.vertex_offset;
        float clip[4];
        vertex_stage(g, inst, f.proj_view, vi, P[k], clip);
        quat_rotate(inst.rotation, g.normal[vi * 3u], g.normal[vi * 3u + 1u], g.normal[vi * 3u + 2u], N[k]);
        r.T[k][0] = g.uv[vi * 2u];
        r.T[k][1] = g.uv[vi * 2u + 1u];
        X[k] = (clip[0] + clip[3]) * hw;
        Y[k] = (clip[1] + clip[3]) * hh;
        W[k] = clip[3];
        r.z[k] = clip[2];
        r.w[k] = clip[3];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        r.A[i] = Y[k] * W[j] - W[k] * Y[j];
        r.B[i] = W[k] * X[j] - X[k] * W[j];
        r.C[i] = X[k] * Y[j] - Y[k] * X[j];
    }
    const float det = (X[0] * r.A[0] + Y[0] * r.B[0]) + W[0] * r.C[0];
    const bool front = det > 0.0f;
    int x0 = 0, y0 = 0, x1 = (int)f.width - 1, y1 = (int)f.height - 1;
    if (W[0] > 0.0f && W[1] > 0.0f && W[2] > 0.0f) {
        const float xs0 = X[0] / W[0], xs1 = X[1] / W[1], xs2 = X[2] / W[2];
        const float ys0 = Y[0] / W[0], ys1 = Y[1] / W[1], ys2 = Y[2] / W[2];
        const float lim = 16777216.0f;
        const float xmin = fmaxf(fminf(fminf(xs0, fminf(xs1, xs2)), lim), -lim), xmax = fmaxf(fminf(fmaxf(xs0, fmaxf(xs1, xs2)), lim), -lim);
        const float ymin = fmaxf(fminf(fminf(ys0, fminf(ys1, ys2)), lim), -lim), ymax = fmaxf(fminf(fmaxf(ys0, fmaxf(ys1, ys2)), lim), -lim);
        x0 = max(x0, (int)floorf(xmin) - 1);
        y0 = max(y0, (int)floorf(ymin) - 1);
        x1 = min(x1, (int)floorf(xmax) + 1);
        y1 = min(y1, (int)floorf(ymax) + 1);
    }
    uint32_t items = 0;
    if (front && x0 <= x1 && y0 <= y1) {
        r.x0 = (uint16_t)x0; r.y0 = (uint16_t)y0; r.x1 = (uint16_t)x1; r.y1 = (uint16_t)y1;
        const uint32_t rows = (uint32_t)(y1 >> 3) - (uint32_t)(y0 >> 3) + 1u;
        const uint32_t cols = (uint32_t)(x1 >> 3) - (uint32_t)(x0 >> 3) + 1u;
        items = rows * ((cols + kItemWidthBlocks - 1u) / kItemWidthBlocks);
    } else {
        r.x0 = 1; r.x1 = 0; r.y0 = 1; r.y1 = 0;
    }
    // the class of the material, for the tile coverage words the shading launches steer by (shade_kernel's TEX launches)
    const uint32_t cls = material_flags ? ((material_flags[(size_t)inst.material_id * flags_stride] & 12u) == 4u ? 2u : 4u) : 6u;
    r.flags = ((second && alpha_buffer_b) ? 1u : 0u) | cls;
    r.material_id = inst.material_id;
    for (int k = 0; k < 7; ++k) r._pad[k] = 0u;
    records[t] = r;
    tr_tri_planes pl;   // what the shading launches / the resolve interpolate from (tr_visibility.h)
    tri_planes_from_record(r, P, N, inst.translation_and_scale[3], pl);
    tri_planes[t] = pl;
    return items;
}

// item_base = exclusive prefix of the work-item counts over the layer's triangles, formed INSIDE the set-up launch by a
// decoupled look-back (Merrill & Garland's single-pass scan): a workgroup scans its 256 counts, publishes its aggregate in
// a status word and adds up its predecessors' words until it meets one that already holds an inclusive prefix.  The words
// are 64-bit agent-scope atomics that CARRY the value — (epoch << 34 | state << 32 | value) — so no fence is needed (a
// device-scope release writes back the calling XCD's whole L2 on this chip: DESIGN.md), and the epoch (the context's frame
// counter) makes last frame's words read as "not yet": nothing is cleared between frames.  This replaces the separate
// scan launches (one ~5 us launch for small layers, three for large ones).
constexpr unsigned long long kScanAggregate = 1ull, kScanInclusive = 2ull;
__device__ __forceinline__ unsigned long long scan_word(uint32_t epoch, unsigned long long state, uint32_t value) {
    return ((unsigned long long)epoch << 34) | (state << 32) | (unsigned long long)value;
}
// Called by the first wave of workgroup `block` (all 64 lanes); returns the exclusive prefix of the block.
__device__ __forceinline__ uint32_t scan_lookback(unsigned long long* __restrict__ status, uint32_t block, uint32_t aggregate,
                                                  uint32_t epoch, uint32_t lane) {
    if (lane == 0u)
        __hip_atomic_store(&status[block], scan_word(epoch, block == 0u ? kScanInclusive : kScanAggregate, aggregate), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (block == 0u) return 0u;
    uint32_t exclusive = 0u;
    int base = (int)block - 1;
    for (;;) {
        const int idx = base - (int)lane;                       // lane k looks at predecessor base - k
        unsigned long long w = scan_word(epoch, kScanInclusive, 0u);   // (before block 0: prefix 0)
        uint64_t pending, inclusive;
        for (;;) {
            if (idx >= 0) w = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool current = (uint32_t)(w >> 34) == epoch;
            const uint32_t state = current ? (uint32_t)(w >> 32) & 3u : 0u;
            pending = ballot(state == 0u);
            inclusive = ballot(state == (uint32_t)kScanInclusive);
            // usable when, walking back from the nearest predecessor, an inclusive word comes before any missing one
            const int first_missing = pending ? __ffsll((unsigned long long)pending) - 1 : 64;
            const int first_inclusive = inclusive ? __ffsll((unsigned long long)inclusive) - 1 : 64;
            if (first_inclusive < first_missing || first_missing == 64) break;
            __builtin_amdgcn_s_sleep(2);
        }
        const int stop = inclusive ? __ffsll((unsigned long long)inclusive) - 1 : 63;   // last lane whose value counts
        uint32_t v = (int)lane <= stop ? (uint32_t)w : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
        exclusive += v;
        if (inclusive) break;
        base -= 64;
    }
    if (lane == 0u)
        __hip_atomic_store(&status[block], scan_word(epoch, kScanInclusive, exclusive + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return exclusive;
}

struct tr_alpha_tables {        // what the alpha-clip kill reads (depth_pre_pass_alpha_clip, shader/src/lib.rs:269-292)
    const tr_material_info* materials;
    const tr_dtex* textures;
    const uint32_t* tex_arena;
    uint32_t num_textures;
};

// Both layers in one launch (blockIdx.y = layer): the transmissive layer's fragments are NOT tested against the opaque
// depth here — its winner is the nearest transmissive fragment of the pixel, and whether that one lies in front of the
// opaque surface is decided by whoever consumes the opaque word last (the resolve, or the opaque VIS launch: it zeroes a
// hidden transmissive word).  The result is the same — if the nearest transmissive fragment is hidden, all are; if not,
// it is also the nearest of the visible ones — and the small layer's launch (19 us of mostly latency on the demo frame)
// runs beside the large one's instead of behind it.
struct tr_raster_layers {
    const tr_tri_record* records[2];
    const uint32_t* item_base[2];
    const tr_layer_counts* counts[2];
    unsigned long long* vis[2];
    uint32_t* tile_cover[2];     // [ceil(h/4)][ceil(w/64)], zeroed
    uint32_t enabled[2];         // (a layer that cannot have triangles has no buffers)
};
// Measurement hooks of the rasteriser: the product build defines them away; tools/build_variant.py NAME -DTR_RASTER_TIMING=1
// gets them from tr_raster_probe.h (per-wave phase counters and a per-wave log of the opaque layer's launch).
#ifdef TR_RASTER_TIMING
#include "tr_raster_probe.h"
#else
#define TR_RT(x)
#define TR_RT_WAVE_BEGIN
#define TR_RT_WAVE_END
#endif
// The part of tr_tri_record the coverage and depth tests read (tr_visibility.h: the words before T), wave-uniform in
// scalar registers.
constexpr uint32_t kRasterRecordWords = 20u;
static_assert(offsetof(tr_tri_record, T) == 76 && offsetof(tr_tri_record, flags) == 68, "raster_record follows tr_tri_record's layout");
struct raster_record {
    float A[3], B[3], C[3], z[3], w[3];
    uint32_t x0, y0, x1, y1, flags;
};
__global__ __launch_bounds__(256) void raster_kernel(const tr_geometry_view g, const tr_raster_frame f, const tr_raster_layers rl,
                                                     const tr_alpha_tables alpha, uint32_t* __restrict__ scan_epoch) {
#pragma clang fp contract(off)
    // The frame counter the set-up launch in front of this one tagged its look-back words with (scan_lookback) moves on here,
    // on the device: a captured frame replays with a new tag every time.  30 bits, never 0 (= never written).
    if (blockIdx.x == 0u && blockIdx.y == 0u && threadIdx.x == 0u) {
        const uint32_t e = __hip_atomic_load(scan_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(scan_epoch, e >= 0x3FFFFFFFu ? 1u : e + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t layer = blockIdx.y;
    if (!rl.enabled[layer]) return;
    const tr_tri_record* __restrict__ records = rl.records[layer];
    const uint32_t* __restrict__ item_base = rl.item_base[layer];
    const tr_layer_counts* __restrict__ counts = rl.counts[layer];
    unsigned long long* __restrict__ vis = rl.vis[layer];
    uint32_t* __restrict__ tile_cover = rl.tile_cover[layer];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * 4u;
    const uint32_t n_items = counts->num_items, n_tris = counts->num_triangles;
    const uint32_t first = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));   // wave-uniform
    // Wave w owns the items w, w + W, w + 2W, ... (W = number of waves: neighbouring rows of a large triangle go to
    // different waves).  It takes 64 of them at a time: every lane finds the triangle of one item (a binary search
    // over the prefix array, 64 searches in flight together), then the wave works through the 64 items one by one.
    TR_RT_WAVE_BEGIN
    // one or two batches per wave: the search's round trips are exposed (the waves all search together), and the wide search
    // is 1.7 us of the 4K demo frame; with many batches the waves hide each other's trips and its sevenfold loads cost 4 %
    const bool few_batches = n_items <= 128u * waves;
    for (uint32_t batch = 0; first + (uint64_t)batch * 64u * waves < n_items; ++batch) {
      TR_RT(const unsigned long long rt_t0 = TR_RT_NOW();)
      const uint64_t my_item64 = first + ((uint64_t)batch * 64u + lane) * waves;
      const uint32_t my_item = (uint32_t)min(my_item64, (uint64_t)0xFFFFFFFFu);
      // Every lane prepares one item: the triangle (a binary search over the prefix array, 64 searches in flight
      // together), what the rasteriser reads of its record (the first 76 bytes, into registers of the lane), the item's
      // place in the triangle's bounds, and which of its blocks can hold a fragment at all.  Half of the blocks in a
      // large triangle's bounds miss it, and in the bounds of a long thin one nearly all do: an edge function, evaluated
      // as the pixel test evaluates it, is monotone in x and in y (every rounding is), so its largest value over a
      // block's pixel centres is the one at the corner chosen by the signs of A and B; negative there = no pixel of the
      // block is inside.  Exactly the blocks the per-pixel test would find empty or a superset are visited: the result
      // is unchanged.  The wave then works through the items that have a block left, one by one — an item costs it 20
      // v_readlane or so; the division, the block tests and the empty items (85 % of them in a scene of large triangles at a
      // grazing angle: tools/gpu_raster_stress.py) cost one lane's work each, 64 at a time.
      uint32_t my_t = 0u, my_place = 0u, my_blocks = 0u;     // my_place = block row | first block << 16; my_blocks bit b: block first + b
      const bool mine = my_item64 < n_items;
      uint32_t my_rec[kRasterRecordWords] = {};
      if (mine) {
          my_t = few_batches ? upper_index_wide(item_base, n_tris, my_item) : upper_index(item_base, n_tris, my_item);
          const uint4* src = reinterpret_cast<const uint4*>(records + my_t);
#pragma unroll
          for (uint32_t q = 0; q < kRasterRecordWords / 4u; ++q) {
              const uint4 v = src[q];
              my_rec[q * 4u] = v.x; my_rec[q * 4u + 1u] = v.y; my_rec[q * 4u + 2u] = v.z; my_rec[q * 4u + 3u] = v.w;
          }
          const uint32_t local = my_item - item_base[my_t];
          const uint32_t bx0 = (my_rec[15] & 0xFFFFu) >> 3, by0 = my_rec[15] >> 19, bx1 = (my_rec[16] & 0xFFFFu) >> 3;
          const uint32_t groups = ((bx1 - bx0 + 1u) + kItemWidthBlocks - 1u) / kItemWidthBlocks;
          const uint32_t row = local / groups, group = local - row * groups;
          const uint32_t by = by0 + row;
          const uint32_t bstart = bx0 + group * kItemWidthBlocks, bend = min(bstart + kItemWidthBlocks - 1u, bx1);
          uint32_t blocks = (2u << (bend - bstart)) - 1u;   // bit b: block bstart + b
          const float yl = (float)(by * 8u) + 0.5f;
#pragma unroll
          for (uint32_t b = 0; b < kItemWidthBlocks; ++b) {
              const float xl = (float)((bstart + b) * 8u) + 0.5f;
              bool maybe = true;
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                  const float A = __uint_as_float(my_rec[i]), B = __uint_as_float(my_rec[3 + i]), C = __uint_as_float(my_rec[6 + i]);
                  const float xc = A > 0.0f ? xl + 7.0f : xl, yc = B > 0.0f ? yl + 7.0f : yl;
                  maybe &= (A * xc + B * yc) + C >= 0.0f;
              }
              if (!maybe) blocks &= ~(1u << b);
          }
          my_place = by | (bstart << 16);
          my_blocks = blocks;
      }
      unsigned long long live = ballot(my_blocks != 0u);
      TR_RT(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); rt_search += TR_RT_NOW() - rt_t0;)
      while (live) {
        TR_RT(const unsigned long long rt_t1 = TR_RT_NOW(); ++rt_items;)
        const int k = __ffsll((unsigned long long)live) - 1;
        live &= live - 1ull;
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)my_t, k);
        const uint32_t place = (uint32_t)__builtin_amdgcn_readlane((int)my_place, k);
        raster_record rec;
        {
            uint32_t words[18];
#pragma unroll
            for (uint32_t q = 0; q < 18u; ++q) words[q] = (uint32_t)__builtin_amdgcn_readlane((int)my_rec[q], k);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                rec.A[i] = __uint_as_float(words[i]);      rec.B[i] = __uint_as_float(words[3 + i]);
                rec.C[i] = __uint_as_float(words[6 + i]);  rec.z[i] = __uint_as_float(words[9 + i]);
                rec.w[i] = __uint_as_float(words[12 + i]);
            }
            rec.x0 = words[15] & 0xFFFFu; rec.y0 = words[15] >> 16; rec.x1 = words[16] & 0xFFFFu; rec.y1 = words[16] >> 16;
            rec.flags = words[17];
        }
        const uint32_t by = place & 0xFFFFu, bstart = place >> 16;
        uint32_t blocks = (uint32_t)__builtin_amdgcn_readlane((int)my_blocks, k);
        const uint32_t py = by * 8u + (lane >> 3);
        const bool alpha_clip = (rec.flags & 1u) != 0u;
        // an alpha-clipped draw's item reads what its kill needs once, not once per block (the material, then the texture's
        // descriptor: two of a block's four dependent round trips — the level offsets and the texels are the other two; such a
        // block takes 4x the time of a plain one by the probe build's counters, and the waves that draw them end the launch)
        float kill_factor = 1.0f, kill_cutoff = 0.0f, kill_T[3][2] = {};
        int32_t kill_tex = -1;
        tr_dtex_head kill_head = {};
        const TR_CONSTANT uint32_t* kill_offsets = nullptr;
        if (alpha_clip) {
            const TR_CONSTANT tr_tri_record& whole = *as_constant(records + t);   // (not in `rec`: the material and the uv corners)
            const TR_CONSTANT tr_material_info& m = *as_constant(alpha.materials + whole.material_id);
            kill_factor = m.diffuse_factor[3];
            kill_cutoff = m.alpha_clipping_cutoff;
            kill_tex = m.textures.diffuse;
#pragma unroll
            for (int i = 0; i < 3; ++i) { kill_T[i][0] = whole.T[i][0]; kill_T[i][1] = whole.T[i][1]; }
            if (kill_tex >= 0 && (uint32_t)kill_tex < alpha.num_textures) {
                cdtex* d = as_constant(alpha.textures) + kill_tex;
                kill_head = texture_head(d);
                kill_offsets = d->offset;
            }
        }
        // the item's fragments tag the coverage words of the (at most two) 64-pixel tile columns it spans, upper and lower
        // half: collected here (scalar) and written once behind the block loop instead of once per block
        uint32_t cover_bits[kItemTileColumns][2] = {};
        TR_RT(const unsigned long long rt_t2 = TR_RT_NOW(); rt_pro += rt_t2 - rt_t1;)
        while (blocks) {
            TR_RT(++rt_nblocks; rt_alpha += alpha_clip ? 1ull : 0ull;)
            const uint32_t bx = bstart + (uint32_t)(__ffs((int)blocks) - 1);
            blocks &= blocks - 1u;
            const uint32_t px = bx * 8u + (lane & 7u);
            float fv[3], lam[3], depth;
            bool hit = tri_edges(rec, (float)px + 0.5f, (float)py + 0.5f, fv);
            hit = hit && px >= rec.x0 && px <= rec.x1 && py >= rec.y0 && py <= rec.y1;
            if (ballot(hit) == 0ull) continue;   // the block misses the triangle (half of a large triangle's box does)
            hit = tri_depth(rec, fv, lam, depth) && hit;
            const size_t pix = (size_t)py * f.width + px;
            if (hit && alpha_clip) {
                // implicit-LOD fetch of the diffuse texture: uv at the two quad partners from the same triangle
                // (what helper invocations compute), differences oriented like dFdx / dFdy
                float alpha_v = kill_factor;
                if (kill_offsets != nullptr) {
                    auto uv_at = [&](const float l[3], int c) { return (l[0] * kill_T[0][c] + l[1] * kill_T[1][c]) + l[2] * kill_T[2][c]; };
                    float lx[3], ly[3], dd;
                    tri_pixel(rec, (float)(px ^ 1u) + 0.5f, (float)py + 0.5f, lx, dd);
                    tri_pixel(rec, (float)px + 0.5f, (float)(py ^ 1u) + 0.5f, ly, dd);
                    const float u = uv_at(lam, 0), v = uv_at(lam, 1);
                    const float sx = (px & 1u) ? -1.0f : 1.0f, sy = (py & 1u) ? -1.0f : 1.0f;
                    uv_derivs dv;
                    dv.dudx = (uv_at(lx, 0) - u) * sx;
                    dv.dvdx = (uv_at(lx, 1) - v) * sx;
                    dv.dudy = (uv_at(ly, 0) - u) * sy;
                    dv.dvdy = (uv_at(ly, 1) - v) * sy;
                    texture_fetch tf;
                    texture_issue(tf, alpha.tex_arena, kill_head, kill_offsets, u, v, dv);
                    alpha_v *= texture_resolve_channel<3>(tf, false, nullptr);
                } else if (kill_tex != -1) {
                    alpha_v = 0.0f;   // unbound slot reads as zero
                }
                hit = !(alpha_v < kill_cutoff);
            }
            // (Testing the word with a plain or agent-scope load first and skipping fragments that already lose — the word only
            //  grows within a frame — measured SLOWER, 57.6 -> 62.5 us on the 4K mesh frame: the load costs an L2 channel slot
            //  like the atomic it saves, and an instruction is saved only when all 64 lanes lose.)
            if (hit) atomicMax(&vis[pix], ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned long long)t);
            // the 64x4 block tiles that received a fragment (the 8x8 block lies in two of them, one above the other), and
            // the material classes of what landed there (bit 0 = touched, bit 1 = full-class textured, bit 2 = any other;
            // conservative: a fragment that loses the depth test later has tagged its tile all the same).  The resolve
            // and the shading launches skip tiles by these words.
            const unsigned long long hits = ballot(hit);
            TR_RT(rt_frags += (unsigned long long)__popcll(hits);)
            const uint32_t col = (bx >> 3) - (bstart >> 3), bits = 1u | (rec.flags & 6u);
            if ((uint32_t)hits != 0u) cover_bits[col][0] |= bits;
            if ((uint32_t)(hits >> 32) != 0u) cover_bits[col][1] |= bits;
        }
        TR_RT(rt_blocks += TR_RT_NOW() - rt_t2;)
        if (lane == 0u) {
            const uint32_t cover_w = (f.width + 63u) >> 6;
            uint32_t* c = tile_cover + (size_t)(by * 2u) * cover_w + (bstart >> 3);
#pragma unroll
            for (uint32_t col = 0; col < kItemTileColumns; ++col) {
                if (cover_bits[col][0] != 0u) atomicOr(c + col, cover_bits[col][0]);
                if (cover_bits[col][1] != 0u) atomicOr(c + col + cover_w, cover_bits[col][1]);
            }
        }
      }
    }
    TR_RT_WAVE_END
}

struct tr_layer_planes {
    float4* pos_depth;
    float4* nrm_scale;
    float2* uv;
    uint32_t* material_id;
};

// ---- both layers per launch ----------------------------------------------------------------------------------------
// The front end of a layer (scan of the draw stream, vertex stage + setup, the three scan passes over the work items)
// is five small launches, each bound by launch latency at scene sizes like the demos'; the two layers are
// independent until the transmissive layer's coverage test reads the opaque depth, so each of the five runs once
// with the layer in blockIdx.y (the resolve: blockIdx.z), on work buffers of its own.
struct tr_layer_work {
    const tr_draw_command* draws_a;
    const tr_draw_command* draws_b;
    uint32_t buffer_a, capacity_triangles;     // capacity 0: the layer cannot have triangles, its front end is skipped
    uint32_t* tri_base;
    tr_layer_counts* counts;
    tr_tri_record* records;
    tr_tri_planes* tri_planes;
    uint32_t* item_base;
    unsigned long long* vis;
    tr_layer_planes planes;
    uint32_t* tile_cover;                      // [ceil(h/4)][ceil(w/64)]: zeroed per frame, set by raster_kernel
};
struct tr_two_layers {
    tr_layer_work l[2];
};
#define TR_PICK_LAYER(two, which) const tr_layer_work W = (which) ? (two).l[1] : (two).l[0]

// One thread per pixel and BOTH layers: the winning triangles' attributes at the pixel centre (vis_interpolate), or "no
// fragment".  The transmissive layer's winner counts only if it is nearer than the opaque surface (see raster_kernel).
__device__ __forceinline__ void raster_resolve_layer(const tr_raster_frame f, const tr_tri_planes* __restrict__ tri_planes,
                                                     unsigned long long key, const tr_layer_planes out, size_t pix, uint32_t px, uint32_t py) {
    if (key == 0ull) {   // no fragment: only the id plane is defined there (the shading passes look at nothing else)
        out.material_id[pix] = TR_NOT_COVERED;
        return;
    }
    vis_fragment v;
    vis_planes_interpolate(tri_planes[(uint32_t)key], key, px, py, v);
    out.pos_depth[pix] = float4{v.position[0], v.position[1], v.position[2], v.depth};
    out.nrm_scale[pix] = float4{v.normal[0], v.normal[1], v.normal[2], v.scale};
    out.uv[pix] = float2{v.uv[0], v.uv[1]};
    out.material_id[pix] = v.material_id;
}
__device__ __forceinline__ void raster_resolve_body(const tr_raster_frame f, const tr_two_layers& two, bool ids_of_untouched_tiles) {
    const uint32_t px = blockIdx.x * 64u + (threadIdx.x & 63u), py = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (px >= f.width || py >= f.height) return;
    const size_t pix = (size_t)py * f.width + px;
    // one word per 64x4 block tile (this workgroup) and layer, set by raster_kernel when a fragment landed in it: an
    // untouched tile has no fragment, its visibility words need not be read
    const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
    unsigned long long key[2];
#pragma unroll
    for (uint32_t l = 0; l < 2u; ++l) {
        const bool touched = two.l[l].tile_cover && as_constant(two.l[l].tile_cover)[tile] != 0u;
        key[l] = touched ? two.l[l].vis[pix] : 0ull;
        // The resolve is the last reader of a visibility word: it leaves the buffer zeroed for the next frame, so a frame
        // clears only the words it set instead of filling both whole-frame buffers (133 MB at 4K, 21 us) up front.
        if (key[l] != 0ull) two.l[l].vis[pix] = 0ull;
    }
    if (key[1] != 0ull && !(__uint_as_float((uint32_t)(key[1] >> 32)) > __uint_as_float((uint32_t)(key[0] >> 32)))) key[1] = 0ull;
#pragma unroll
    for (uint32_t l = 0; l < 2u; ++l) {
        const bool touched = two.l[l].tile_cover && as_constant(two.l[l].tile_cover)[tile] != 0u;
        if (!touched && !ids_of_untouched_tiles) continue;
        raster_resolve_layer(f, two.l[l].tri_planes, key[l], two.l[l].planes, pix, px, py);
    }
}



__global__ __launch_bounds__(1024) void raster_scan_draws_kernel(const tr_two_layers two, const uint32_t* __restrict__ draw_counts,
                                                                 uint32_t capacity_draws) {
    TR_PICK_LAYER(two, blockIdx.y);
    if (W.capacity_triangles == 0u) return;
    raster_scan_draws_body(W.draws_a, W.draws_b, draw_counts, W.buffer_a, capacity_draws, W.capacity_triangles, W.tri_base, W.counts);
}
// Vertex stage + triangle set-up + the work-item prefix of both layers (blockIdx.y = layer).  The grid is sized for the
// layer's capacity; the triangle count lives on the device: workgroups past it exit, the one that holds the last triangle
// closes the prefix.
__global__ __launch_bounds__(256) void raster_setup_kernel(const tr_geometry_view g, const tr_raster_frame f, const tr_two_layers two,
                                                           const uint32_t* __restrict__ material_flags, uint32_t flags_stride,
                                                           unsigned long long* __restrict__ scan_status, uint32_t status_stride,
                                                           const uint32_t* __restrict__ scan_epoch) {
    // this frame's tag, advanced by the rasteriser launch behind this one.  (An agent-scope load, past the scalar and vector
    // caches: replayed from a HIP graph, a plain or scalar load here returned the PREVIOUS replay's value — the look-back
    // then read that replay's words as current)
    // (requested here, read where the look-back starts: made scalar at once, it is a memory round trip in front of everything else)
    const uint32_t epoch_word = __hip_atomic_load(scan_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TR_PICK_LAYER(two, blockIdx.y);
    if (W.capacity_triangles == 0u) return;
    __shared__ uint32_t lds_wave[4];
    __shared__ uint32_t lds_prefix;
    const uint32_t n = W.counts->num_triangles;
    const uint32_t first = blockIdx.x * 256u, t = first + threadIdx.x;
    if (first >= n) {
        if (n == 0u && blockIdx.x == 0u && threadIdx.x == 0u) {
            W.item_base[0] = 0u;
            W.counts->num_items = 0u;
        }
        return;
    }
    const uint32_t items = t < n ? raster_setup_body(g, f, W.draws_a, W.draws_b, W.tri_base, W.counts, 1u, W.records, W.tri_planes,
                                                     material_flags, flags_stride, t)
                                 : 0u;
    // exclusive scan over the workgroup's 256 counts
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = items;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    if (lane == 63u) lds_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0u, total = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        before += k < wave ? lds_wave[k] : 0u;
        total += lds_wave[k];
    }
    if (wave == 0u) {
        const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)epoch_word);
        const uint32_t ex = scan_lookback(scan_status + (size_t)blockIdx.y * status_stride, blockIdx.x, total, epoch, lane);
        if (lane == 0u) lds_prefix = ex;
    }
    __syncthreads();
    const uint32_t prefix = lds_prefix;
    if (t < n) W.item_base[t] = prefix + before + incl - items;
    if (first + 256u >= n && threadIdx.x == 0u) {   // the workgroup of the last triangle closes the prefix
        W.item_base[n] = prefix + total;
        W.counts->num_items = prefix + total;
    }
}
__global__ __launch_bounds__(256) void raster_resolve_kernel(const tr_raster_frame f, const tr_two_layers two,
                                                             uint32_t ids_of_untouched_tiles) {
    raster_resolve_body(f, two, ids_of_untouched_tiles != 0u);
}

}  // namespace tr
