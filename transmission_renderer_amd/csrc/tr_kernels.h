// tr_kernels.h — gfx950 (CDNA4, wave64) device code of the transmission/volume PBR shading path.
//
// Shape of the work
//   One thread shades one pixel; a wave is a 16x4 pixel tile (four 256 B row segments per float4 plane, one 128 B
//   segment of the RGBA16F target per row) and a workgroup of its own, so the dispatcher refills a wave slot the
//   moment its wave retires.  The 64x4 block tiles of the rect are numbered so that each XCD (own L2) owns a
//   contiguous band of the screen (or, for real frames, stripes of it): the data-dependent taps into the opaque
//   pyramid then re-use texel rows inside one L2 instead of being fetched by all eight.
//
// What bounds it (rocprofv3 PMC and the TR_ABLATION builds): with the sun and one light HBM — the pass
// moves its 52 B per pixel at 0.9 of the rate its own streaming skeleton reaches, and switching the lights off changes
// nothing — because the ~410 vector instructions per 64-pixel tile hide behind the stream; every further light is vector
// work on top.  That is why the math is arranged to be short, NOT as a transliteration of the reference, and why the
// kernel is kept at 64 VGPRs (8 waves per SIMD): resident waves are what overlaps the arithmetic with the scalar-load,
// G-buffer and tap latencies:
//   * everything that depends only on the material is digested once per upload into `tr_dmat`
//     and read through the scalar unit: a wave handles one material at a time (waves that
//     straddle several run a waterfall loop over them), so there is a single code path and the
//     material costs no vector registers;
//   * per light, the halfway-vector algebra collapses onto n.l, v.l and |n x (v+l)|^2:
//       |v+l|^2 = 2+2 v.l,  v.h = (1+v.l)/|v+l|,  1-(n.h)^2 = |n x (v+l)|^2/|v+l|^2
//     (the last form keeps d_ggx well conditioned at low roughness); the mirrored light of
//     transmission_btdf needs no vector at all (n.l' = -n.l, v.l' = v.l - 2 n.l n.v, and
//     n x (v+l') = n x (v+l)), so the two lobes share the vector work;  D*V is one reciprocal per lobe,
//     x^5 three multiplies.  (On gfx950 v_pk_fma_f32 issues at half the rate of v_fma_f32 — tools/ubench —
//     so packing buys nothing: the build disables packed-fp32 selection.)
//   * horizontally adjacent texels come in one 16-byte load, filtered as one weighted sum of v_fma_mix_f32;
//   * the cluster x / y lookups are exact tables (built with the reference's own IEEE division on the host); the
//     depth slice is one v_log_f32 made exact against a table of slice thresholds (depth_slice);
//   * the light list is walked on the scalar unit: one list per tile in the usual case, a waterfall over the tile's
//     distinct clusters otherwise (cluster_lookup);
//   * every buffer is addressed as scalar base + 32-bit byte offset (saddr/voffset loads, 24-bit multiply-adds);
//   * the pyramid / LUT taps are issued after the light loop (see shade_pixel) and there is no register
//     prefetch of the next tile: both were worth less than the two extra waves their registers cost.
// These differ from the reference's op order by a few ulp of fp32 (and are closer to exact
// arithmetic where the reference is ill-conditioned); parity criteria: tests/test_gpu_parity.py.
// Variants that were measured and rejected are recorded in DESIGN.md 3.1 (and in the history of this file), not kept
// here as switches.
//
// Reference semantics implemented here (file:line relative to the reference root):
//   fragment_transmission        shader/src/lib.rs:37-162
//   fragment                     shader/src/lib.rs:164-249
//   evaluate_lights[_transmission] shader/src/lighting.rs:13-95, 145-220
//   basic_brdf / transmission_btdf / ibl_volume_refraction   glam-pbr/src/lib.rs:377-423, 200-233, 292-354
//   cluster lookup               shader/src/lib.rs:88-98, shared-structs/src/lib.rs:54-63
#pragma once

#include "tr_common.h"
#include "tr_texture_kernels.h"
#include "tr_visibility.h"

// Measurement hooks.  The product build defines every one of them away; profiling builds (tools/build_variant.py NAME
// -DTR_ABLATION=1 / -DTR_TIMING=1 / -DTR_PROBE_MASK=n — never the product library, __graft_entry__.compile_library refuses)
// get them from tr_probe.h: phases that can be switched off (TR_ABLATE) and per-wave wait-cycle counters (TR_PROBE_*).
#if defined(TR_ABLATION) || defined(TR_TIMING) || defined(TR_PROBE_MASK)
#include "tr_probe.h"
#else
#define TR_ABLATION 0
#define TR_ABLATE(L, bit) false
#define TR_PROBE_ARGS_DECL
#define TR_PROBE_ARGS
#define TR_PROBE_WAVE_BEGIN
#define TR_PROBE_SINCE(name)
#define TR_PROBE_DRAIN
#define TR_PROBE_WAITED(slot, name)
#define TR_PROBE_TILE_DONE
#define TR_PROBE_WAVE_END
#endif

namespace tr {

// Issue priority rises through the phases of a tile: among the waves of a SIMD the one nearest to the end of its tile
// wins the arbitration, finishes, and has its next tile's loads in flight while the others compute — without it the
// round-robin arbitration keeps waves that started together in lockstep (all wait, then all compute).
template <int P>
__device__ __forceinline__ void tile_phase() {
    __builtin_amdgcn_s_setprio(P);
}
constexpr uint32_t kFrontLists = 64u;      // sub-lists of the transmissive-covered tile list (tr_launch::front_list)
constexpr uint32_t kStripeTileRows = 1u;   // VIS / textured launches: tile rows per XCD stripe (round 4: 1 ... 8 measured the same; with background
                                           // tiles written from constants 1 / 2 / 4 / 8 / 16 -> 4K mesh frame 163.3 / 163.6 / 164.8 / 167.5 / 172.8 us;
                                           // fractions of a row are no better, and 15 tiles — every XCD the same columns — 173)
constexpr uint32_t kParkedValues = 17u;    // full-class textured pixels: values parked in LDS, see shade_pixel_textured

// ---------------------------------------------------------------- digested material (240 B)
// Index 0 of every pair belongs to the basic_brdf lobe, index 1 to the transmission_btdf lobe.
struct alignas(16) tr_dmat {
    float diffuse[3];      // diffuse_factor.rgb (base colour)
    float f90;             // calculate_combined_f90 (a splat)
    float c_diff[3];       // lerp(diffuse, 0, metallic) * (1/pi)
    float eta;             // 1 / ior
    float f0[3];           // calculate_combined_f0
    float transmission_factor;
    float df[3];           // f90 - f0
    float thickness;
    float emission[3];
    float rough_ior;       // roughness * clamp(2 ior - 2, 0, 1)   (pyramid lod = log2(W) * this)
    float neg_atten_log2[3];  // -(-ln(colour)/distance) * log2(e); 0 when distance == +INF
    float lut_fy;          // GGX LUT row interpolation weight   (v = perceptual roughness)
    float a2[2];           // alpha^2: (roughness^2)^2 ; (roughness^2 * clamp(2 ior - 2, 0, 1))^2
    float oma2[2];         // 1 - a2
    float k[2];            // a2 * 0.5 / pi   (numerator of D*V)
    uint32_t lut_row0;     // GGX LUT row offsets into the pair table (entries)
    uint32_t lut_row1;
    uint32_t flags;        // bit0: finite attenuation distance; bit1: transmission_factor != 0;
                           // bit2: the material has texture slots (shaded by the per-pixel material path);
                           // bit3: ... of the lite class (lite_dmat);
                           // bit4: kd is zero in every channel (transmission_factor == 1 or metallic == 1): the
                           //       transmissive pass needs no diffuse sum; bit5: c_diff is (metallic == 1): nor the opaque;
                           // bit6: the btdf lobe's constants bt_a, bt_b are +-0 in every channel (alpha_t = 0: ior 1, or roughness 0):
                           //       its sums are multiplied by zero in the resolve and need not be formed
    float ior_clamp;       // clamp(2 ior - 2, 0, 1)
    float f0_dielectric;   // ((ior - 1) / (ior + 1))^2
    float bt_a[3];         // k[1] * (1 - f0): the btdf lobe is accumulated as sum(I D'V') and sum(I D'V' p') and
    float bt_b[3];         // k[1] * (f90 - f0): resolved once per pixel as bt_a * sum1 - bt_b * sum2
    uint32_t lut_line;     // first entry of this material's GGX LUT line (build_lut_lines_kernel), in entries
    float neg_eta2;        // -(eta * eta)   (refract's k = 1 - eta^2 (1 - (n.i)^2))
    uint32_t _pad;
    // The composite of a pixel whose material has no texture slots (shade_pixel's finish), with tf = transmission_factor
    // folded in: lib.rs:157-159 is  diffuse' = lerp(D, tf * X, tf) = (1 - tf) D + tf^2 X  with D = c_diff * sum_d and
    // X = diffuse * ((1 - (f0 A + f90 B)) T + bt_a * sum_ta - bt_b * sum_tb), so
    //   out = kd * sum_d + kt * (1 - (f0 A + f90 B)) T + kta * sum_ta - ktb * sum_tb + specular + emission
    float kd[3];           // c_diff * (1 - tf)
    float omf0_max;        // 1 - max(f0): the channel whose Fresnel term is the largest at every angle, and
    float kt[3];           // tf^2 * diffuse
    float ndf_min;         //   -(f90 - max(f0)):  1 - max(F) = omf0_max + ndf_min * p
    float kta[3];          // tf^2 * diffuse * bt_a
    float _pad3;
    float ktb[3];          // tf^2 * diffuse * bt_b
    float _pad4;
    // The basic_brdf lobe is accumulated without its constant k[0] = a2 / 2 pi, like the btdf lobe (bt_a, bt_b): the
    // per-pixel resolve  specular = f0 * sum(I nol D*V) + (f90 - f0) * sum(I nol D*V p)  reads k[0] folded into both
    float ks_f0[3];        // k[0] * f0
    float _pad5;
    float ks_df[3];        // k[0] * (f90 - f0)
    float _pad6;
};
static_assert(sizeof(tr_dmat) == 272, "digested material is 272 B");

// Light as the kernels read it: the reference's 48-byte record (shared-structs/src/lib.rs:70-78)
// with the per-light constants of spotlight_factor (:129-138) digested at upload.
struct alignas(16) tr_dlight {
    float pos[3];      float inv_spot_epsilon;  // 1 / (cos(inner) - cos(outer))
    float colour[3];   uint32_t is_spot;        // spotlight_direction_and_outer_angle.w != 0
    float spot_dir[3]; float cos_outer;         // cos(outer_angle)
};
static_assert(sizeof(tr_dlight) == 48, "Light is 48 B");

struct tr_level_table {           // pyramid geometry, one entry per mip level
    uint32_t offset[TR_MAX_MIP_LEVELS];  // texels from pyramid base
    uint32_t width[TR_MAX_MIP_LEVELS];
    uint32_t height[TR_MAX_MIP_LEVELS];
    float wf[TR_MAX_MIP_LEVELS];         // (float)width, (float)height
    float hf[TR_MAX_MIP_LEVELS];
    float xlim[TR_MAX_MIP_LEVELS];       // max(width - 2, 0): left texel of the right-most 16-byte pair
};

// What the refraction tap of a material WITHOUT texture slots needs from the opaque pyramid (digest_taps_kernel, once
// per (material table, pyramid geometry, framebuffer width)): the pyramid lod of ibl_volume_refraction
// (glam-pbr/src/lib.rs:334-335) depends on the material only, so its level pair, the interpolation weight and the
// geometry of the two levels are one scalar record per material — the pixel neither forms the lod nor walks the level
// table (eight vector instructions on wave-uniform values and a dependent scalar round trip per tile before).
// Index 0 of every pair is the lower level (floor(lod)), index 1 the next.
struct alignas(16) tr_dtap {
    float t;                  // frac(lod): weight of level 1
    uint32_t narrow;          // bit k: level k is one texel wide (the 16-byte pair's second texel is not its neighbour)
    uint32_t offset[2];       // byte offset of the level from the pyramid base
    uint32_t pitch[2];        // bytes from a row to the next; 0 when the level has one row (both taps read that row)
    uint32_t width[2];        // texels per row
    float wf[2], hf[2];       // (float)width, (float)height
    float xhi[2], yhi[2];     // width - 1, height - 1: the clamp of the texel coordinate
    float xlim[2], ylim[2];   // max(width - 2, 0), max(height - 2, 0): the first texel / row of the last pair
    uint32_t level[2];        // the two levels' indices (a row-band sharded pass checks its taps of levels 0 and 1 against
    uint32_t _pad[2];         //   the rows it holds: tap_window_excess)
};
static_assert(sizeof(tr_dtap) == 96, "tap record is 96 B");

// Everything a shading launch needs besides the planes; passed by value (kernarg -> SGPRs).
struct tr_frame_params {
    float proj_view[16];
    float view_position[3];
    float log2_fb_width;         // log2(framebuffer_size.x as f32)
    float sun_dir[3];
    float slice_k;               // depth slice ~ slice_k - lcc_scale * log2(slice_fpn - (2 (1 - depth) - 1) * slice_fmn),
    float sun_intensity[3];      //   made exact against the slice_thr table (depth_slice)
    float lcc_scale;
    float slice_fpn, slice_fmn;  // z_far + z_near, z_far - z_near (fp32, as the reference forms them)
    uint32_t slice_max;          // get_depth_slice(+0.0): the last slice a depth in [0, inf] can fall into
    uint32_t clusters_xy;        // num_clusters.x * num_clusters.y
    uint32_t num_clusters_total;
    uint32_t debug_clusters;
    uint32_t width, height;      // frame size (colour-target pitch)
    uint32_t g_width;            // G-buffer plane pitch
    uint32_t g_origin_x, g_origin_y;  // frame position of plane element (0,0)
    uint32_t rect_x0, rect_y0, rect_x1, rect_y1;
    uint32_t tiles_x, tiles_y;   // 64x4 tiles covering the rect
    uint32_t tiles_x_magic;      // floor(2^32 / tiles_x): tile / tiles_x on the scalar unit (one fix-up step)
    uint32_t j_step;             // waves of the grid per XCD (per sub-list when the launch walks the front list): what a wave's tile
                                 // index advances by — from the host: read through the hidden grid-size argument, its pointer
                                 // occupies a scalar register pair across the whole tile
    uint32_t stripe_tiles;       // VIS launches: block tiles per stripe of kStripeTileRows tile rows, and
    uint32_t stripe_magic;       // floor(2^32 / stripe_tiles)
    // Rank-interleaved strips (tr_set_strips; 0 tile rows = off): the rect is the frame, tile row r of the launch is the
    // frame's tile row ((r / T) * world + rank) * T + r % T — strip k of this rank is strip k * world + rank of the frame
    uint32_t strip_tile_rows, strip_magic /* floor(2^32 / T) */, strip_world, strip_rank;
    // Row-band sharded full pipeline (tr_set_tap_window; hi = 0: off): the rows [lo, hi) of pyramid level 0 this rank
    // holds — its own band and the halo it received — and, halved, of level 1; levels >= 2 are whole on every rank.
    uint32_t tap_row_lo, tap_row_hi;
    float lut_wf;                // (float)lut_width
    uint32_t lut_stride;         // pair-table stride in entries (= lut_width + 2)
    uint32_t lut_height;
    uint32_t pyr_levels;
    uint32_t ablate;             // profiling builds only (tr_probe.h): which phases are switched off; 0 in the product
};

typedef const TR_CONSTANT tr_dmat cdmat;
typedef const TR_CONSTANT tr_dlight cdlight;
typedef const TR_CONSTANT tr_level_table clevels;
typedef const TR_CONSTANT tr_dtap cdtap;
typedef const TR_CONSTANT uint32_t cu32;

// The single kernel argument of shade_kernel: frame parameters and every pointer.  The kernel reads it
// through the kernarg segment pointer, re-"laundered" at the start of each phase (`launder`): a value
// is then fetched by s_load in the phase that uses it instead of being loaded in the prologue and kept
// (or spilled to VGPR lanes — ~100 v_readlane/v_writelane per tile before this) across the whole kernel.
struct tr_launch {
    tr_frame_params fp;
    const tr_dmat* dmats;
    const tr_dlight* lights;
    const uint32_t* cluster_counts;
    const uint32_t* light_indices;
    const uint32_t* lut_pairs;          // (R,G)[x-1], (R,G)[x] per entry
    const float4* lut_lines;            // per material: the LUT at the material's roughness, (A,B)[x-1], (A,B)[x] per entry
    const tr_level_table* levels;
    const tr_dtap* dtaps;               // per material, see tr_dtap
    const uint32_t* cluster_x;          // [frame width]  u32(frag_coord.x / cluster_size.x)
    const uint32_t* cluster_y_term;     // [frame height] u32(frag_coord.y / cluster_size.y) * num_clusters.x
    const float4* pos_depth;
    const float4* nrm_scale;
    const uint32_t* material_id;
    const uint2* pyramid;
    void* hdr;
    uint2* mip0;
    uint2* mip1;                        // optional (opaque launches of the frame recorder, even frame sizes): level 1 of the
                                        // opaque pyramid, written from the 2x2 quads of the values this launch stores
    // textured materials only (shade_kernel<.., TEXTURED = true>)
    const float2* uv;
    const tr_material_info* materials;  // the raw records: factors and texture ids
    const struct tr_dtex* textures;
    const uint32_t* tex_arena;          // RGBA8 texels of every chain
    const float* srgb_to_linear;        // 256 entries
    const float* slice_thr;             // [slice_max + 2] depth thresholds of get_depth_slice, see depth_slice()
    const uint32_t* tile_cover;         // optional: one word per 64x4 block tile of the frame, 0 = the layer has no fragment there
                                        // (bit 1 / bit 2: fragments of a full-class material / of any other, raster_kernel)
    // VIS launches (the frame recorder): the layer's visibility words and triangle records instead of the planes
    unsigned long long* vis;
    const tr_tri_planes* tri_planes;
    // opaque VIS launches: the transmissive layer's words and coverage map.  The rasteriser keeps the pixel's NEAREST
    // transmissive fragment whatever lies in front of it (raster_kernel); the opaque launch, the last to know the opaque
    // depth, zeroes that word where the fragment is not nearer than the opaque surface.
    unsigned long long* vis_front;
    const uint32_t* cover_front;
    uint32_t* front_list_build;         // (see front_list below)
    uint32_t* front_list_build_count;
    uint32_t* tap_excess;               // optional (tap window set): atomicMax of the level-0 rows a tap reached beyond the window
    // VIS launches of a frame that is also presented (tr_record_frame with a tonemap target): the launch that writes a
    // pixel's FINAL colour — the transmissive one, or the opaque one where no transmissive winner survives — tonemaps
    // the RGBA16F value it stores (fragment_tonemap on the same bits) into `present`: the frame has no tonemap pass
    // that reads the whole target back (99 MB of traffic at 4K, 17 us at the memory roofline)
    // The transmissive layer usually covers a fraction of the screen.  The opaque VIS launch, which visits every block tile
    // and reads the layer's coverage word anyway, lists the tiles that hold transmissive fragments (`front_list_build`, one
    // atomic per such tile); the transmissive VIS launch walks that list (`front_list`) instead of the frame: dense work
    // for its waves instead of five tiles each of which 70 % are skipped.
    const uint32_t* front_list;
    const uint32_t* front_list_count;
    uint32_t front_list_cap;            // entries per sub-list: the list is kFrontLists lists with a counter each (tile t goes into list
                                        // t % kFrontLists: one counter would be a queue of same-address atomics, 11-14 ns apart —
                                        // thousands of tiles per frame); the walking launch's workgroup b walks list b % kFrontLists
    uint32_t* present;
    tr_tonemap_params present_params;
    float present_e1;
    int32_t present_bgra;
};
typedef const TR_CONSTANT tr_launch claunch;

// The material as the per-pixel code reads it when it has texture slots: the same fields as tr_dmat, but
// per lane (vector registers), digested per pixel from the sampled factors.
struct lane_dmat {
    float diffuse[3], f90, eta, f0[3], transmission_factor, thickness, emission[3], ior_clamp;
    float neg_atten_log2[3], a2[2];
    // (1 - a2, a2 / 2 pi, f90 - f0 and roughness * ior_clamp are one instruction each where they are used: keeping them
    //  would cost eight more registers across the whole pixel — the difference between four and five waves per SIMD)
    float metallic, rough;   // what the end of the pixel derives c_diff, the btdf coefficients and the LUT row from
    uint32_t flags;          // (kept instead of those values: they are needed only after the light loop)
};
__device__ __forceinline__ const lane_dmat* launder(const lane_dmat* p) { return p; }

// Values read only after the light loop: from the table for a scalar material, derived on the spot for a per-lane one.
__device__ __forceinline__ float mat_c_diff(const TR_CONSTANT tr_dmat* m, int k) { return m->c_diff[k]; }
__device__ __forceinline__ float mat_bt_a(const TR_CONSTANT tr_dmat* m, int k) { return m->bt_a[k]; }
__device__ __forceinline__ float mat_bt_b(const TR_CONSTANT tr_dmat* m, int k) { return m->bt_b[k]; }
__device__ __forceinline__ float mat_c_diff(const lane_dmat* m, int k) {
    const float diff = m->diffuse[k];
    return (diff + (0.0f - diff) * m->metallic) * kFrac1Pi;          // lerp(diffuse, 0, metallic) / pi
}
// the derived constants: table fields of a scalar record, recomputed for a per-lane one
__device__ __forceinline__ float m_oma2(const TR_CONSTANT tr_dmat& m, int k) { return m.oma2[k]; }
__device__ __forceinline__ float m_k(const TR_CONSTANT tr_dmat& m, int k) { return m.k[k]; }
__device__ __forceinline__ float m_df(const TR_CONSTANT tr_dmat& m, int k) { return m.df[k]; }
__device__ __forceinline__ float m_rough_ior(const TR_CONSTANT tr_dmat& m) { return m.rough_ior; }
__device__ __forceinline__ float m_neg_eta2(const TR_CONSTANT tr_dmat& m) { return m.neg_eta2; }
// (the operand goes through an empty volatile asm: the recomputation stays where it is used instead of being hoisted
//  out of the light loop back into a register of its own)
__device__ __forceinline__ float here(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float m_oma2(const lane_dmat& m, int k) { return 1.0f - here(m.a2[k]); }
__device__ __forceinline__ float m_k(const lane_dmat& m, int k) { return here(m.a2[k]) * (0.5f * kFrac1Pi); }
__device__ __forceinline__ float m_df(const lane_dmat& m, int k) { return m.f90 - here(m.f0[k]); }
__device__ __forceinline__ float m_rough_ior(const lane_dmat& m) { return m.rough * m.ior_clamp; }
__device__ __forceinline__ float m_neg_eta2(const lane_dmat& m) { return -m.eta * m.eta; }
__device__ __forceinline__ float mat_bt_a(const lane_dmat* m, int k) { return m_k(*m, 1) * (1.0f - m->f0[k]); }
__device__ __forceinline__ float mat_bt_b(const lane_dmat* m, int k) { return m_k(*m, 1) * m_df(*m, k); }

// The "lite" material class: a dielectric (metallic_factor == 0) whose only bound texture is the base colour — the most
// common textured glTF material.  Everything the specular lobes read stays what the material table holds (f0 does not
// depend on the base colour when metallic is 0), so the pixel keeps the scalar record and carries just the sampled base
// colour per lane: three registers instead of the per-lane record's thirty.
struct lite_dmat {
    cdmat* m;
    float diffuse[3];      // diffuse_factor.rgb * sample (lib.rs:65-69)
};
__device__ __forceinline__ const lite_dmat* launder(const lite_dmat* p) { return p; }
// the record the material-constant reads go to
__device__ __forceinline__ cdmat* mat_base(cdmat* m) { return m; }
__device__ __forceinline__ const lane_dmat* mat_base(const lane_dmat* m) { return m; }
__device__ __forceinline__ cdmat* mat_base(const lite_dmat* m) { return launder(m->m); }
// the values a base-colour texture changes
__device__ __forceinline__ float mat_diffuse(const TR_CONSTANT tr_dmat* m, int k) { return m->diffuse[k]; }
__device__ __forceinline__ float mat_diffuse(const lane_dmat* m, int k) { return m->diffuse[k]; }
__device__ __forceinline__ float mat_diffuse(const lite_dmat* m, int k) { return m->diffuse[k]; }
__device__ __forceinline__ float mat_c_diff(const lite_dmat* m, int k) {
    const float diff = m->diffuse[k];
    return (diff + (0.0f - diff) * 0.0f) * kFrac1Pi;                 // lerp(diffuse, 0, metallic = 0) / pi
}

// min / max against a table value in a scalar register, as ONE instruction: for a value it cannot prove canonical
// (anything loaded) the compiler puts a canonicalising v_max x, x in front of fminf / fmaxf.  NaN in `a` gives `s`.
__device__ __forceinline__ float min_s(float a, float s) {
    float r;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(a));
    return r;
}
__device__ __forceinline__ float max_v(float a, float b) {   // fmaxf of two register values, no canonicalisation
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max(x, eps): Dot::new's clamp (glam-pbr/src/lib.rs:93-98) as one instruction (the literal is the first source)
__device__ __forceinline__ float clamp_eps(float x) {
    float r;
    asm("v_max_f32_e32 %0, 0x34000000, %1" : "=v"(r) : "v"(x));
    return r;
}

// -max(-x, eps) = min(x, -eps): the clamped dot product of the mirrored light with its sign left on (one instruction; the
// negation is a free source modifier where the value is used)
__device__ __forceinline__ float neg_clamp_eps(float x) {
    float r;
    asm("v_min_f32_e32 %0, 0xb4000000, %1" : "=v"(r) : "v"(x));
    return r;
}
// 1 - max(a * b, eps) = min(1 - a * b, 1 - eps)  (1 - 2^-23 is a float): one fused multiply-add and one min instead of
// multiply, max, subtract; NaN gives 1 - eps on both sides
__device__ __forceinline__ float one_minus_clamped(float a, float b) {
    float r;
    const float t = fmaf(-a, b, 1.0f);
    asm("v_min_f32_e32 %0, 0x3f7ffffe, %1" : "=v"(r) : "v"(t));
    return r;
}
// ------------------------------------------------------------------------ one light
// Accumulators of one pixel over its lights.
struct light_acc {
    f3 d;        // sum I * nol * (1 - max(F))                     (x c_diff/pi at the end)
    f3 s;        // sum I * nol * D*V * F                          (basic_brdf specular) — for a material in scalar registers
    f3 sp;       //   F = f0 + (f90 - f0) p is resolved once per pixel like the btdf lobe: s = sum I nol D*V, sp = sum I nol D*V p
    f3 ta, tb;   // sum I * D'V'/k' and sum I * D'V'/k' * p'       (transmission_btdf: (1 - F') is linear in p', so the
                 //  lobe is resolved once per pixel: bt_a * ta - bt_b * tb; the per-light work has no scalar operand)
};

// One light against one pixel: basic_brdf (+ transmission_btdf when TRANSMISSIVE), the two lobes
// evaluated side by side in the halves of packed fp32 instructions.
//   n, v unit; l unit direction to the light; I = rgb intensity reaching the pixel.
//
//   f of d_ggx (glam-pbr/src/lib.rs:101-109): the reference evaluates f = noh^2 (a2 - 1) + 1 with
//   noh = n.h; at low roughness (a2 ~ 1e-6) that needs 1 - noh^2 to ~1e-9 absolute, which the
//   fp32 form only delivers by luck of rounding.  Here 1 - noh^2 = sin2 = |n x (v+l)|^2 / |v+l|^2
//   has fp32 *relative* accuracy and f = sin2 + a2 (1 - sin2) has no cancellation.  noh is
//   clamped to EPSILON by Dot::new when n.h <= 0 (:93-98): then f = 1 + eps^2 (a2 - 1) = 1.
//   |n x (v+l)|^2 is shared by both lobes: the mirrored light l' = l - 2 (n.l) n differs from l by
//   a multiple of n, and n x n = 0.
// Per-pixel, light-independent inputs of eval_light.
struct pixel_frame {
    f3 n, v;           // unit normal and view vector
    v2f nvx, nvy, nvz; // { n , v } component pairs: n.l and v.l come out of three packed ops
    float nov_raw, nov;  // n.v and max(n.v, EPSILON)
    v2f g_nov;         // sqrt(n.v^2 (1 - a2) + a2) per lobe: the light-independent half of v_smith
};

// FIRST: the pixel's first light (the sun) writes the accumulators instead of adding to zeros — `0 + x` is not `x`
// for the compiler (signed zeros), so accumulating into zero-initialised registers costs a move and an add per sum.
template <bool TRANSMISSIVE, bool FIRST = false, class Mat /* cdmat (scalar registers) or const lane_dmat (per lane) */>
__device__ __forceinline__ void eval_light(light_acc& acc, Mat& m, const pixel_frame& px, f3 l, f3 I, bool btdf,
                                           bool diffuse = true /* (scalar) acc.d is wanted: see tr_dmat::flags bit 4 */) {
    constexpr bool SPLIT_F = std::is_same<Mat, cdmat>::value;
    const f3 n = px.n, v = px.v;
    const float nov_raw = px.nov_raw, nov = px.nov;
    const float a2_0 = m.a2[0], a2_1 = m.a2[1];
    const float nl_raw = dot3(n.x, n.y, n.z, l.x, l.y, l.z);
    const float vl = dot3(v.x, v.y, v.z, l.x, l.y, l.z);
    const float hx = v.x + l.x, hy = v.y + l.y, hz = v.z + l.z;
    const float cx = fmaf(n.y, hz, -(n.z * hy)), cy = fmaf(n.z, hx, -(n.x * hz)), cz = fmaf(n.x, hy, -(n.y * hx));
    const float c2 = dot3(cx, cy, cz, cx, cy, cz);

    // ---- lobe 0: basic_brdf (glam-pbr/src/lib.rs:377-423)
    // |v+l|^2 = 2 + 2 v.l (|v| = |l| = 1) vanishes where v = -l; rounding can push it below zero: floor it.
    {
        const float inv_h = rsq(fmaxf(fmaf(2.0f, vl, 2.0f), 1e-12f));
        const float omv = one_minus_clamped(1.0f + vl, inv_h);     // 1 - v.h, v.h clamped to EPSILON by Dot::new (:93-98)
        const float nol = clamp_eps(nl_raw);
        const float omv2 = omv * omv, p = omv2 * omv2 * omv;       // fresnel_schlick :137-139
        const float sin2 = c2 * (inv_h * inv_h);                   // 1 - (n.h)^2
        const float f = (nov_raw + nl_raw) > 0.0f ? fmaf(a2_0, 1.0f - sin2, sin2) : 1.0f;
        // v_smith_ggx_correlated (:114-133) and D*V with one reciprocal
        // (nol^2 (1 - a2) + a2 written as nol^2 + a2 (1 - nol^2): one table operand per instruction — with two the
        //  compiler first copies one into a vector register, and an instruction with a scalar operand never pairs)
        const float nol2 = nol * nol;
        const float g = fmaf(nol, px.g_nov.x, nov * fast_sqrt(fmaf(a2_0, 1.0f - nol2, nol2)));
        // (g > 0 always: n.l, n.v are clamped to EPSILON and the roots are positive, so v_smith's `denom <= 0`
        //  guard, :125-131, cannot trigger; a NaN propagates like in the reference)
        // (a scalar-record material's k[0] is folded into the constants of the per-pixel resolve: tr_dmat::ks_f0)
        const float dv = SPLIT_F ? rcp(f * f * g) : m_k(m, 0) * rcp(f * f * g);
        const float ws = nol * dv;                                 // specular_brdf :362-375 (weighted by n.l :414-421)
        if constexpr (SPLIT_F) {
            // max(F) = F of the channel with the largest f0 (f90 is a splat and p <= 1: F is monotone in f0)
            // (scalar branch: a material whose diffuse constant is zero — transmission_factor 1, or a metal — never reads
            //  the diffuse sum: six instructions per light, four of them with a scalar operand)
            if (diffuse) {
                const float wd = nol * fmaf(m.ndf_min, p, m.omf0_max); // diffuse_brdf :356-360: 1 - max(F)
                if constexpr (FIRST) {
                    acc.d = {I.x * wd, I.y * wd, I.z * wd};
                } else {
                    acc.d.x = fmaf(I.x, wd, acc.d.x);
                    acc.d.y = fmaf(I.y, wd, acc.d.y);
                    acc.d.z = fmaf(I.z, wd, acc.d.z);
                }
            }
            const float wsp = ws * p;
            if constexpr (FIRST) {
                acc.s = {I.x * ws, I.y * ws, I.z * ws};
                acc.sp = {I.x * wsp, I.y * wsp, I.z * wsp};
            } else {
                acc.s.x = fmaf(I.x, ws, acc.s.x);
                acc.s.y = fmaf(I.y, ws, acc.s.y);
                acc.s.z = fmaf(I.z, ws, acc.s.z);
                acc.sp.x = fmaf(I.x, wsp, acc.sp.x);
                acc.sp.y = fmaf(I.y, wsp, acc.sp.y);
                acc.sp.z = fmaf(I.z, wsp, acc.sp.z);
            }
        } else {
            const float Fx = fmaf(m_df(m, 0), p, m.f0[0]), Fy = fmaf(m_df(m, 1), p, m.f0[1]), Fz = fmaf(m_df(m, 2), p, m.f0[2]);
            const float wd = nol * (1.0f - fmaxf(Fx, fmaxf(Fy, Fz)));  // diffuse_brdf :356-360
            if constexpr (FIRST) {
                acc.d = {I.x * wd, I.y * wd, I.z * wd};
                acc.s = {I.x * ws * Fx, I.y * ws * Fy, I.z * ws * Fz};
            } else {
                acc.d.x = fmaf(I.x, wd, acc.d.x);
                acc.d.y = fmaf(I.y, wd, acc.d.y);
                acc.d.z = fmaf(I.z, wd, acc.d.z);
                acc.s.x = fmaf(I.x * ws, Fx, acc.s.x);
                acc.s.y = fmaf(I.y * ws, Fy, acc.s.y);
                acc.s.z = fmaf(I.z * ws, Fz, acc.s.z);
            }
        }
    }
    // ---- lobe 1: transmission_btdf (:200-233): the light mirrored about the surface,
    //      n.l' = -(n.l), v.l' = v.l - 2 (n.l)(n.v); no vector is formed.  For the mirrored light |v+l'|^2
    //      vanishes exactly at the specular peak of lobe 0 (v + l' = (v+l) - 2 (n.l) n = 0 when h = n), so the
    //      floor is hit on real pixels (the reference normalises a ~1e-8 vector there and gets an arbitrary but
    //      finite h; F' = f90 makes the lobe's weight vanish either way).
    //      Skipped (scalar branch) when the material's transmission_factor is 0: lib.rs:157-159 then multiplies
    //      everything this lobe feeds by zero.
    if constexpr (TRANSMISSIVE) {
        if (btdf) {
            const float vlm = fmaf(-2.0f * nl_raw, nov_raw, vl);
            const float inv_h = rsq(fmaxf(fmaf(2.0f, vlm, 2.0f), 1e-12f));
            const float omv = one_minus_clamped(vlm + 1.0f, inv_h);
            const float neg_nolm = neg_clamp_eps(nl_raw);            // -max(-(n.l), EPSILON): the sign rides on its uses
            const float omv2 = omv * omv, p = omv2 * omv2 * omv;
            const float sin2 = c2 * (inv_h * inv_h);
            const float f = (nov_raw - nl_raw) > 0.0f ? fmaf(a2_1, 1.0f - sin2, sin2) : 1.0f;
            const float nolm2 = neg_nolm * neg_nolm;
            const float g = fmaf(-neg_nolm, px.g_nov.y, nov * fast_sqrt(fmaf(a2_1, 1.0f - nolm2, nolm2)));
            const float r = rcp(f * f * g);                          // D'V' / k[1]; not weighted by n.l (:232)
            if constexpr (FIRST) {
                const float tx = I.x * r, ty = I.y * r, tz = I.z * r;
                acc.ta = {tx, ty, tz};
                acc.tb = {tx * p, ty * p, tz * p};
            } else {
                const float rp = r * p;
                acc.ta.x = fmaf(I.x, r, acc.ta.x);
                acc.ta.y = fmaf(I.y, r, acc.ta.y);
                acc.ta.z = fmaf(I.z, r, acc.ta.z);
                acc.tb.x = fmaf(I.x, rp, acc.tb.x);
                acc.tb.y = fmaf(I.y, rp, acc.tb.y);
                acc.tb.z = fmaf(I.z, rp, acc.tb.z);
            }
        } else if constexpr (FIRST) {
            acc.ta = acc.tb = {0.f, 0.f, 0.f};
        }
    } else if constexpr (FIRST) {
        acc.ta = acc.tb = {0.f, 0.f, 0.f};
    }
}

template <bool TRANSMISSIVE, class Mat>
__device__ __forceinline__ void eval_punctual(light_acc& acc, Mat& m, cdlight& L, f3 pos, const pixel_frame& px,
                                              bool btdf, bool diffuse = true) {
    // light_direction_and_attenuation (glam-pbr/src/lib.rs:12-23): bare 1/d^2
    float dx = L.pos[0] - pos.x, dy = L.pos[1] - pos.y, dz = L.pos[2] - pos.z;
    float inv_d = rsq(dot3(dx, dy, dz, dx, dy, dz));
    f3 l = {dx * inv_d, dy * inv_d, dz * inv_d};
    float att = inv_d * inv_d;
    if constexpr (!TRANSMISSIVE) {
        // Light::spotlight_factor (shared-structs/src/lib.rs:129-138); shader/src/lighting.rs:201-203
        // applies it in `fragment` only — the transmissive loop (:58-92) has no spotlight factor.
        if (L.is_spot) {
            float theta = -dot3(l.x, l.y, l.z, L.spot_dir[0], L.spot_dir[1], L.spot_dir[2]);
            att *= fmaxf((theta - L.cos_outer) * L.inv_spot_epsilon, 0.0f);
        }
    }
    f3 I = {L.colour[0] * att, L.colour[1] * att, L.colour[2] * att};
    eval_light<TRANSMISSIVE, false>(acc, m, px, l, I, btdf, diffuse);
}

// ------------------------------------------------------------------ opaque pyramid taps
// clamp_sampler (src/main.rs:694-705): LINEAR min/mag, LINEAR mip, CLAMP_TO_EDGE.  Per axis:
//   x = u*w - 0.5;  xc = clamp(x, 0, w-1);  i0 = floor(xc), i1 = min(i0+1, w-1), weight xc - i0
// (identical to clamping floor(x) and floor(x)+1 to the edge, and defined for NaN/inf u).  The
// taps (i0, i1) of a row are fetched as one 16-byte load of texels (b, b+1), b = min(i0, w-2): at
// the right edge b = w-2 and the weight xc - b becomes exactly 1.  Levels narrower than 2 texels
// take single-texel loads.
struct tap_pair {   // two mip levels side by side
    v2f wx, wy;     // weights
    uint32_t a00[2], a01[2];  // texel index of (row0, b) and (row1, b) per level
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));  // two adjacent RGBA16F texels, texel (8-byte) aligned

__device__ __forceinline__ float h2f_lo(uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu))); }
__device__ __forceinline__ float h2f_hi(uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w >> 16))); }

// rgb of the bilinear taps of both levels -> trilinear result.
struct pyramid_fetch {
    uint4 r0[2], r1[2];   // per level: row0 pair, row1 pair
    v2f wx, wy;
    float t;
    bool narrow0, narrow1;  // (scalar) the level is one texel wide: the pair's second texel is not its neighbour
};

__device__ __forceinline__ void axis_pair(float u, v2f dimf, v2f limf, v2f& w, uint32_t (&b)[2]) {
    v2f x = pk_fma(splat(u), dimf, splat(-0.5f));
    v2f hi = dimf - 1.0f;
    v2f xc = {fminf(fmaxf(x.x, 0.0f), hi.x), fminf(fmaxf(x.y, 0.0f), hi.y)};
    v2f bf = {fminf(floorf(xc.x), limf.x), fminf(floorf(xc.y), limf.y)};  // limf = max(w - 2, 0)
    w = xc - bf;
    b[0] = (uint32_t)bf.x;
    b[1] = (uint32_t)bf.y;
}

__device__ __forceinline__ void axis_single(float u, float dimf, float& w, uint32_t& i0, uint32_t& i1) {
    float x = fmaf(u, dimf, -0.5f);
    float xc = fminf(fmaxf(x, 0.0f), dimf - 1.0f);
    float fl = floorf(xc);
    w = xc - fl;
    i0 = (uint32_t)fl;
    i1 = (uint32_t)fminf(fl + 1.0f, dimf - 1.0f);
}

// Sharded full pipeline: how many level-0 rows the bilinear rows (row, row + 1) of `level` lie outside the window of rows
// this rank holds ([lo, hi) of level 0, [lo / 2, hi / 2) of level 1; 0 = inside, or a level every rank holds whole).
__device__ __forceinline__ float tap_window_excess(uint32_t lo, uint32_t hi, uint32_t level, float row) {
    if (level >= 2u) return 0.0f;
    const float first = (float)(lo >> level), last = (float)((hi >> level) - 1u);
    return fmaxf(fmaxf(first - row, (row + 1.0f) - last), 0.0f) * (float)(1u << level);
}
__device__ __forceinline__ void tap_window_report(uint32_t* excess_word, float excess) {
    if (excess > 0.0f) atomicMax(excess_word, (uint32_t)excess + 1u);   // (rare: a fallback frame follows)
}

// Issues the loads of framebuffer.sample_by_lod(clamp_sampler, uv, lod) (shader/src/lib.rs:135-138) for the
// lanes whose lower level is the (scalar) `l0`, so level geometry is scalar.  Every row of
// every level is one 16-byte load of two adjacent texels; in a level that is a single texel wide the
// second one belongs to the next row / level (or to the 8 bytes of tail padding tr_pyramid_layout
// reserves) and is replaced by the first before use.
__device__ __forceinline__ void pyramid_issue_levels(pyramid_fetch& pf, const uint2* __restrict__ texels,
                                                     clevels* lv, uint32_t levels, float u, float v, uint32_t l0,
                                                     uint32_t win_lo = 0u, uint32_t win_hi = 0u, uint32_t* excess_word = nullptr) {
    const uint32_t l1 = min(l0 + 1u, levels - 1u);
    const uint32_t w0 = lv->width[l0], w1 = lv->width[l1];
    const uint2* b0 = texels + lv->offset[l0];   // (scalar) level bases
    const uint2* b1 = texels + lv->offset[l1];
    uint32_t bx[2];
    axis_pair(u, v2f{lv->wf[l0], lv->wf[l1]}, v2f{lv->xlim[l0], lv->xlim[l1]}, pf.wx, bx);
    float wy0, wy1;
    uint32_t y00, y01, y10, y11;
    axis_single(v, lv->hf[l0], wy0, y00, y01);
    axis_single(v, lv->hf[l1], wy1, y10, y11);
    pf.wy = v2f{wy0, wy1};
    if (win_hi != 0u) {   // (scalar branch: a row-band sharded pass)
        tap_window_report(excess_word, tap_window_excess(win_lo, win_hi, l0, (float)min(y00, lv->height[l0] >= 2u ? lv->height[l0] - 2u : 0u)));
        tap_window_report(excess_word, tap_window_excess(win_lo, win_hi, l1, (float)min(y10, lv->height[l1] >= 2u ? lv->height[l1] - 2u : 0u)));
    }
    auto ld2 = [](const uint2* level, uint32_t texel) {
        u32x4 t = ld<u32x4_a8>(level, texel * 8u);  // one global_load_dwordx4, saddr + voffset
        return uint4{t.x, t.y, t.z, t.w};
    };
    pf.r0[0] = ld2(b0, mad24(y00, w0, bx[0]));
    pf.r1[0] = ld2(b0, mad24(y01, w0, bx[0]));
    pf.r0[1] = ld2(b1, mad24(y10, w1, bx[1]));
    pf.r1[1] = ld2(b1, mad24(y11, w1, bx[1]));
    pf.narrow0 = w0 < 2u;
    pf.narrow1 = w1 < 2u;
}

// The same taps for a material whose lod is in its tap record (tr_dtap: scalar registers).  Per level and axis:
//   t = clamp(u * size - 0.5, 0, size - 1);  b = min(floor(t), max(size - 2, 0));  weight = t - b
// — in BOTH axes the pair (b, b + 1) is fetched: at the far edge b = size - 2 and the weight becomes exactly 1, which
// is the clamped tap (i0 = i1 = size - 1, any weight) of the sampler.  So the second row is the first row's address
// plus the level's pitch: one offset per level, two scalar bases.  A level of one row has pitch 0, a level of one
// column is flagged narrow (pyramid_resolve).
__device__ __forceinline__ void pyramid_issue_record(pyramid_fetch& pf, const uint2* __restrict__ texels, cdtap* tp, float u,
                                                     float v, uint32_t win_lo = 0u, uint32_t win_hi = 0u, uint32_t* excess_word = nullptr) {
    const char* base = reinterpret_cast<const char*>(texels);
    float wx[2], wy[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float tx = __builtin_amdgcn_fmed3f(fmaf(u, tp->wf[k], -0.5f), 0.0f, tp->xhi[k]);
        const float bx = min_s(floorf(tx), tp->xlim[k]);
        wx[k] = tx - bx;
        const float ty = __builtin_amdgcn_fmed3f(fmaf(v, tp->hf[k], -0.5f), 0.0f, tp->yhi[k]);
        const float by = min_s(floorf(ty), tp->ylim[k]);
        wy[k] = ty - by;
        if (win_hi != 0u) tap_window_report(excess_word, tap_window_excess(win_lo, win_hi, tp->level[k], by));   // (scalar branch)
        const uint32_t at = mad24((uint32_t)by, tp->width[k], (uint32_t)bx) * 8u;
        const char* row0 = base + tp->offset[k];
        const char* row1 = row0 + tp->pitch[k];
        const u32x4 a = ld<u32x4_a8>(row0, at), b = ld<u32x4_a8>(row1, at);   // global_load_dwordx4, saddr + voffset
        pf.r0[k] = uint4{a.x, a.y, a.z, a.w};
        pf.r1[k] = uint4{b.x, b.y, b.z, b.w};
    }
    pf.wx = v2f{wx[0], wx[1]};
    pf.wy = v2f{wy[0], wy[1]};
    pf.t = tp->t;
    pf.narrow0 = (tp->narrow & 1u) != 0u;
    pf.narrow1 = (tp->narrow & 2u) != 0u;
}

// Per-lane lod (roughness from a texture): the wave walks the distinct level pairs of its lanes, normally one or two,
// so that level geometry is scalar inside pyramid_issue_levels.
__device__ __forceinline__ void pyramid_issue(pyramid_fetch& pf, const uint2* __restrict__ texels,
                                              clevels* lv, uint32_t levels, float u, float v,
                                              float lod, uint32_t lane, uint32_t win_lo = 0u, uint32_t win_hi = 0u,
                                              uint32_t* excess_word = nullptr) {
    float l = fminf(fmaxf(lod, 0.0f), (float)(levels - 1u));
    float lf = floorf(l);
    pf.t = l - lf;
    const uint32_t mine = (uint32_t)lf;
    uint64_t todo = ballot(true);
    while (todo) {
        const int first = __ffsll((unsigned long long)todo) - 1;
        const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)mine, first);
        const uint64_t group = ballot(mine == l0);
        todo &= ~group;
        if ((group >> lane) & 1ull) pyramid_issue_levels(pf, texels, lv, levels, u, v, l0, win_lo, win_hi, excess_word);
    }
}

// Filters the fetched texels.  Written as one weighted sum over the 8 taps,
//   sum_level  lw * ( (1-wx)(1-wy) t00 + wx(1-wy) t10 + (1-wx)wy t01 + wx wy t11 ),  lw = (1-t, t),
// so that every term is a v_fma_mix_f32 reading the half-precision texel in place (no v_cvt_f32_f16, which
// runs at half rate, and no subtractions); it equals the lerp form of the oracle up to fp32 rounding.
__device__ __forceinline__ f3 pyramid_resolve(pyramid_fetch& pf) {
    // (scalar, rare: only the last levels of a pyramid are one texel wide; the empty asm keeps the branch a branch —
    //  if-converted, every pixel pays eight v_cndmask for it)
    if (pf.narrow0 || pf.narrow1) {
        asm volatile("");
        if (pf.narrow0) { pf.r0[0].z = pf.r0[0].x; pf.r0[0].w = pf.r0[0].y; pf.r1[0].z = pf.r1[0].x; pf.r1[0].w = pf.r1[0].y; }
        if (pf.narrow1) { pf.r0[1].z = pf.r0[1].x; pf.r0[1].w = pf.r0[1].y; pf.r1[1].z = pf.r1[1].x; pf.r1[1].w = pf.r1[1].y; }
    }
    float w[2][4];
    const float lw[2] = {1.0f - pf.t, pf.t};
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const float wx = l ? pf.wx.y : pf.wx.x, wy = l ? pf.wy.y : pf.wy.x;
        const float top = lw[l] - lw[l] * wy, bot = lw[l] * wy;   // lw (1 - wy), lw wy
        w[l][1] = top * wx;          // t10
        w[l][0] = top - w[l][1];     // t00
        w[l][3] = bot * wx;          // t11
        w[l][2] = bot - w[l][3];     // t01
    }
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        auto ch = [c](const uint4& q, int texel) {
            uint32_t wd = (c < 2) ? (texel ? q.z : q.x) : (texel ? q.w : q.y);
            return (c == 1) ? h2f_hi(wd) : h2f_lo(wd);
        };
        float acc = ch(pf.r0[0], 0) * w[0][0];
        acc = fmaf(ch(pf.r0[0], 1), w[0][1], acc);
        acc = fmaf(ch(pf.r1[0], 0), w[0][2], acc);
        acc = fmaf(ch(pf.r1[0], 1), w[0][3], acc);
        acc = fmaf(ch(pf.r0[1], 0), w[1][0], acc);
        acc = fmaf(ch(pf.r0[1], 1), w[1][1], acc);
        acc = fmaf(ch(pf.r1[1], 0), w[1][2], acc);
        acc = fmaf(ch(pf.r1[1], 1), w[1][3], acc);
        out[c] = acc;
    }
    return {out[0], out[1], out[2]};
}

// textures[ggx_lut].sample(clamp_sampler, (n.v, roughness)).xy (shader/src/lib.rs:126-133).
// The row pair and its weight depend on the material only (tr_dmat); the pair table gives both
// horizontal neighbours of a row in one dword.  R and G ride in the halves of packed ops.
struct lut_fetch {
    uint32_t p0, p1;
    float fx, fy;
    float4 line;   // scalar-material path: both horizontal neighbours of the material's own LUT line
};
__device__ __forceinline__ void lut_issue(lut_fetch& lf, const uint32_t* __restrict__ pairs, float lut_wf,
                                          uint32_t row0, uint32_t row1, float nov_raw) {
    float x = fmaf(nov_raw, lut_wf, -0.5f);
    x = fminf(fmaxf(x, -1.0f), lut_wf);
    float fl = floorf(x);
    lf.fx = x - fl;
    uint32_t k = (uint32_t)((int)fl + 1);
    lf.p0 = ld<uint32_t>(pairs, (row0 + k) * 4u);
    lf.p1 = ld<uint32_t>(pairs, (row1 + k) * 4u);
}
// The same tap for a material in scalar registers: its roughness is fixed, so the row interpolation was done once
// per material (build_lut_lines_kernel) and the pixel only interpolates along n.v: one 16-byte load, four ops.
__device__ __forceinline__ void lut_line_issue(lut_fetch& lf, const float4* __restrict__ lines, float lut_wf,
                                               uint32_t line, float nov_raw) {
    float x = fmaf(nov_raw, lut_wf, -0.5f);
    x = min_s(fmaxf(x, -1.0f), lut_wf);
    float fl = floorf(x);
    lf.fx = x - fl;
    uint32_t k = (uint32_t)((int)fl + 1);
    lf.line = ld<float4>(lines, (line + k) * 16u);
}
__device__ __forceinline__ v2f lut_line_resolve(const lut_fetch& lf) {
    return v2f{fmaf(lf.line.z - lf.line.x, lf.fx, lf.line.x), fmaf(lf.line.w - lf.line.y, lf.fx, lf.line.y)};
}
__device__ __forceinline__ v2f lut_resolve(const lut_fetch& lf, float fy) {
    auto b = [](uint32_t w, int i) { return (float)((w >> (8 * i)) & 0xFFu); };
    v2f t00 = {b(lf.p0, 0), b(lf.p0, 1)}, t10 = {b(lf.p0, 2), b(lf.p0, 3)};
    v2f t01 = {b(lf.p1, 0), b(lf.p1, 1)}, t11 = {b(lf.p1, 2), b(lf.p1, 3)};
    v2f top = pk_fma(t10 - t00, splat(lf.fx), t00);
    v2f bot = pk_fma(t11 - t01, splat(lf.fx), t01);
    return pk_fma(bot - top, splat(fy), top) * (1.0f / 255.0f);
}

// (as a 2-vector of _Float16: gfx950's v_cvt_pk_f16_f32 converts — round to nearest even — and packs a pair in ONE
//  instruction; __floats2half2_rn compiles to two conversions and a v_perm_b32 whose selector sits in a register of its own)
__device__ __forceinline__ uint2 pack_rgba16f(float r, float g, float b, float a) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const half2v lo = {(_Float16)r, (_Float16)g}, hi = {(_Float16)b, (_Float16)a};
    return uint2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
}

// shader/src/lib.rs:647-668
__device__ __forceinline__ f3 debug_colour_for_id(uint32_t id) {
    constexpr float c[15][3] = {{0.0f, 0.0f, 0.0f},      {0.0f, 0.0f, 0.1647f},      {0.0f, 0.0f, 0.3647f},
                                {0.0f, 0.0f, 0.6647f},   {0.0f, 0.0f, 0.9647f},      {0.0f, 0.9255f, 0.9255f},
                                {0.0f, 0.5647f, 0.0f},   {0.0f, 0.7843f, 0.0f},      {1.0f, 1.0f, 0.0f},
                                {0.90588f, 0.75294f, 0.0f}, {1.0f, 0.5647f, 0.0f},   {1.0f, 0.0f, 0.0f},
                                {0.8392f, 0.0f, 0.0f},   {1.0f, 0.0f, 1.0f},         {0.6f, 0.3333f, 0.7882f}};
    uint32_t k = id % 15u;
    return {c[k][0], c[k][1], c[k][2]};
}

// ------------------------------------------------------------------------ material digestion
// The part of the digest that depends on the factors a texture can modulate (metallic, roughness, specular,
// base colour): run once per material at upload for materials without textures (digest_materials_kernel, into
// scalar-read tr_dmat) and once per pixel for textured ones (into lane_dmat).  Same fp32 operations as the
// reference where a value is a pure function of its inputs (glam-pbr/src/lib.rs:141-161, 182-198, 425-435).
// GGX LUT row pair and weight for a roughness (v = perceptual roughness; bilinear, clamp to edge)
__device__ __forceinline__ void lut_rows(float rough, uint32_t lut_height, uint32_t lut_stride, float& fy, uint32_t& row0,
                                         uint32_t& row1) {
#pragma clang fp contract(off)
    const float fh = (float)lut_height;
    float y = rough * fh - 0.5f;
    y = fminf(fmaxf(y, -1.0f), fh);
    const float fl = floorf(y);
    fy = y - fl;
    const int a = (int)fl, mx = (int)lut_height - 1;
    row0 = (uint32_t)min(max(a, 0), mx) * lut_stride;
    row1 = (uint32_t)min(max(a + 1, 0), mx) * lut_stride;
}

template <bool FULL, class D>
__device__ __forceinline__ void digest_factors(D& d, float metallic, float rough, float ior_clamp, float f0d,
                                               float specular_factor, float scx, float scy, float scz, float dx,
                                               float dy, float dz, uint32_t lut_height, uint32_t lut_stride) {
#pragma clang fp contract(off)
    const float alpha = rough * rough;
    const float alpha_t = alpha * ior_clamp;                         // ActualRoughness::apply_ior :144-146
    d.a2[0] = alpha * alpha;
    d.a2[1] = alpha_t * alpha_t;
    if constexpr (FULL) {
        for (int k = 0; k < 2; ++k) {
            d.oma2[k] = 1.0f - d.a2[k];
            d.k[k] = d.a2[k] * (0.5f * kFrac1Pi);
        }
    }
    d.f90 = specular_factor + (1.0f - specular_factor) * metallic;  // calculate_combined_f90
    const float diffuse[3] = {dx, dy, dz}, spec_colour[3] = {scx, scy, scz};
    for (int k = 0; k < 3; ++k) {
        const float diff = diffuse[k];
        d.diffuse[k] = diff;
        const float ds = f0d * spec_colour[k] * specular_factor;
        d.f0[k] = ds + (diff - ds) * metallic;                       // calculate_combined_f0
        if constexpr (FULL) {
            d.df[k] = d.f90 - d.f0[k];
            const float cd = diff + (0.0f - diff) * metallic;        // c_diff = lerp(diffuse, 0, metallic)
            d.c_diff[k] = cd * kFrac1Pi;
            d.bt_a[k] = d.k[1] * (1.0f - d.f0[k]);
            d.bt_b[k] = d.k[1] * d.df[k];
        }
    }
    if constexpr (FULL) d.rough_ior = rough * ior_clamp;             // PerceptualRoughness::apply_ior :157-159
    else d.ior_clamp = ior_clamp;
    if constexpr (FULL) lut_rows(rough, lut_height, lut_stride, d.lut_fy, d.lut_row0, d.lut_row1);
}

// ------------------------------------------------------------------------ cluster lookup
// What a tile knows about its light lists before the light loop.  The list is always walked on the scalar unit: the
// usual tile lies in one cluster (they are 240 pixels wide at 4K and 1/16 of the log-depth range deep), a tile that
// straddles clusters whose lists are the SAME list (count and entries, compared through the scalar unit here) is
// treated like one, and any other tile runs the light loop once per distinct cluster with that cluster's lanes
// (shade_pixel).  There is no per-lane list walk: it cost the pixel five vector registers across the sun and the
// prologue, a second instantiation of the light evaluation and ten register copies where the two walks joined.
constexpr uint32_t kNoCluster = 0xFFFFFFFFu;
struct cluster_list {
    uint32_t cluster;       // cluster index (the debug view prints it)
    uint32_t key;           // the cluster whose list the lane walks; kNoCluster: out of range (robust access: no lights)
    bool uniform;           // (scalar) every pixel of the tile walks the list of s_cluster
    uint32_t s_cluster;     // (scalar) the first working lane's key
    uint32_t s_num, s_l0, s_l1;   // (scalar) its count and the first two entries of its list
};

// LightClusterCoefficients::get_depth_slice (shared-structs/src/lib.rs:43-63), BIT-EXACT for every depth in
// [0, +inf] and NaN (index work).  The reference computes
//     u32(max(log2(2nf / (f + n - (2 (1 - d) - 1)(f - n))) * scale + bias, 0))
// in fp32; that is a composition of monotone correctly-rounded steps, so the slice is a non-increasing step function
// of the depth's bit pattern, and the host (build_slice_thresholds, with the reference's own arithmetic and libm)
// finds by bisection thr[k] = the largest depth whose slice is >= k  (thr[0] = +inf, thr[slice_max + 1] = -1).
// The kernel estimates the slice with one v_log_f32 — the denominator formed by the reference's own roundings, because
// 1 - d loses the low bits of a far depth and the estimate must follow that noise (~3e-3 slices at the far plane) —
// which is within ~1e-5 of the reference's fp32 value; a wave none of whose lanes lies within 2.5e-4 of a slice
// boundary is done, any other fetches the two thresholds around each lane's estimate and corrects it by +-1.
__device__ __forceinline__ float slice_denominator(float depth, float fpn, float fmn) {
#pragma clang fp contract(off)
    const float depth_range = 2.0f * (1.0f - depth) - 1.0f;
    return fpn - depth_range * fmn;
}
__device__ __forceinline__ uint32_t cvt_u32_sat(float x) {   // v_cvt_u32_f32: saturating, NaN -> 0 (Rust `as u32`)
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
constexpr float kSliceGuard = 2.5e-4f;
struct slice_params {   // (scalar registers)
    float scale, fpn, fmn, k;
    uint32_t max;
    const float* thr;
};
__device__ __forceinline__ uint32_t depth_slice(const slice_params& sp, float depth) {
    const float zs = fmaf(-sp.scale, fast_log2(slice_denominator(depth, sp.fpn, sp.fmn)), sp.k);
    uint32_t cz = cvt_u32_sat(zs);
    const float fr = __builtin_amdgcn_fractf(zs);
    if (ballot(fabsf(fr - 0.5f) > 0.5f - kSliceGuard) != 0ull) {   // (rare) a lane near a slice boundary
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        const uint32_t k = min(cz, sp.max);
        const f2u t = ld<f2u>(sp.thr, k * 4u);                      // thr[k], thr[k + 1]
        const uint32_t fixed = k + (depth <= t.y ? 1u : 0u) - (depth > t.x ? 1u : 0u);
        cz = depth >= 0.0f ? fixed : cz;                            // (negative depths: no frag_coord.z is; estimate kept)
    }
    return cz;
}

// (scalar) count of a cluster's list, clamped to the 128 slots (borrowed tables may count past them, like the
// reference's counter); kNoCluster has no lights
__device__ __forceinline__ uint32_t cluster_count(claunch* L, uint32_t sc) {
    if (sc == kNoCluster || TR_ABLATE(L, 8u)) return 0u;
    return min(as_constant(L->cluster_counts)[sc], TR_MAX_LIGHTS_PER_CLUSTER);
}
__device__ __forceinline__ const TR_CONSTANT uint32_t* cluster_entries(claunch* L, uint32_t sc) {
    return as_constant(L->light_indices) + (size_t)(sc == kNoCluster ? 0u : sc) * TR_MAX_LIGHTS_PER_CLUSTER;
}

// shader/src/lib.rs:88-98: x / y from exact tables (cluster_xy), the depth slice from depth_slice().  Runs with the
// whole wave; `has_work`: the lane has a pixel to shade (the others follow the first one that has: a tile with holes
// stays uniform).  At least one lane has work.
__device__ __forceinline__ cluster_list cluster_lookup(claunch* L, float depth, uint32_t cluster_xy, bool has_work) {
    cluster_list c;
    const uint32_t cz = depth_slice(slice_params{L->fp.lcc_scale, L->fp.slice_fpn, L->fp.slice_fmn, L->fp.slice_k,
                                                 L->fp.slice_max, L->slice_thr}, depth);
    c.cluster = mad24(cz, L->fp.clusters_xy, cluster_xy);   // (cz <= slice_max for every depth >= 0: far below 2^24)
    const uint32_t own = c.cluster < L->fp.num_clusters_total ? c.cluster : kNoCluster;
    const uint64_t work = ballot(has_work);
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)own, __ffsll((unsigned long long)work) - 1);
    c.key = has_work ? own : c0;
    uint64_t others = ballot(c.key != c0);
    c.s_cluster = c0;
    // count and the head of the list are requested here, a tile phase ahead of the light loop
    c.s_num = cluster_count(L, c0);
    const TR_CONSTANT uint32_t* list0 = cluster_entries(L, c0);
    c.s_l0 = list0[0];
    c.s_l1 = list0[1];
    bool same = true;
    while (others != 0ull && same) {   // (rare) the tile straddles clusters: is it one list all the same?
        const uint32_t ck = (uint32_t)__builtin_amdgcn_readlane((int)c.key, __ffsll((unsigned long long)others) - 1);
        others &= ~ballot(c.key == ck);
        const TR_CONSTANT uint32_t* list = cluster_entries(L, ck);
        same = cluster_count(L, ck) == c.s_num;
        for (uint32_t i = 0; same && i < c.s_num; ++i) same = list[i] == list0[i];
    }
    c.uniform = same;
    return c;
}

// ------------------------------------------------------------------------ one pixel
// Runs with exec = the lanes of the wave that share material `m` (scalar registers).
// `lane` = lane id in the wave; `cl` = the pixel's cluster and the head of its light list (cluster_lookup).
//
// MatP: `cdmat*` — the digested material in scalar registers (materials without textures), or
// `const lane_dmat*` — per-lane values digested from the sampled textures; `ns.xyz` is then the normal after
// normal mapping.
struct nothing_to_reload {
    __device__ __forceinline__ void operator()() const {}
};
template <bool TRANSMISSIVE, class MatP, class Reload = nothing_to_reload>
__device__ __forceinline__ f3 shade_pixel(claunch* L, MatP m, uint32_t mat_index, float4 pd, float4 ns, uint32_t lane,
                                          const cluster_list& cl TR_PROBE_ARGS_DECL, Reload after_lights = Reload{}) {
    // (roughness, ior and the LUT line are the table's: one level pair and one LUT line per wave)
    constexpr bool SCALAR_MATERIAL = std::is_same<MatP, cdmat*>::value || std::is_same<MatP, const lite_dmat*>::value;
    // ================= phase 1: frame of the pixel, cluster list request, refraction taps =================
    L = launder(L);
    m = launder(m);
    const f3 pos = {pd.x, pd.y, pd.z};
    // view = normalize(view_position - position) (lib.rs:79-80); normal = normalize(n) (lighting.rs:229)
    float vx = L->fp.view_position[0] - pos.x, vy = L->fp.view_position[1] - pos.y, vz = L->fp.view_position[2] - pos.z;
    float inv_v = rsq(dot3(vx, vy, vz, vx, vy, vz));
    const f3 v = {vx * inv_v, vy * inv_v, vz * inv_v};
    float inv_n = rsq(dot3(ns.x, ns.y, ns.z, ns.x, ns.y, ns.z));
    const f3 n = {ns.x * inv_n, ns.y * inv_n, ns.z * inv_n};
    const float nov_raw = dot3(n.x, n.y, n.z, v.x, v.y, v.z);
    const float nov = clamp_eps(nov_raw);
    pixel_frame px;
    px.n = n;
    px.v = v;
    px.nvx = v2f{n.x, v.x};
    px.nvy = v2f{n.y, v.y};
    px.nvz = v2f{n.z, v.z};
    px.nov_raw = nov_raw;
    px.nov = nov;

    uint32_t lights_walked = 0u;   // (opaque pass, debug view) the count of the list the lane walked

    // ---- ibl_volume_refraction, part 1 (glam-pbr/src/lib.rs:292-337): where the refracted ray leaves
    //      the volume, projected to the screen; the taps are in flight while the lights are evaluated.
    pyramid_fetch pf;
    lut_fetch lf;
    float len = 0.0f;
    // transmission_factor == 0 (scalar): lib.rs:157-159 multiplies the whole transmission term by zero, so the
    // refraction taps, the LUT and the btdf lobes are skipped for such materials
    const bool transmits = TRANSMISSIVE && (mat_base(m)->flags & 2u);
    // (scalar) the diffuse sum is read: always for a per-lane or lite record; for a scalar record unless its diffuse
    // constant (kd in the transmissive pass, c_diff in the opaque one) is zero in every channel
    const bool diffuse_on = !std::is_same<MatP, cdmat*>::value || TR_ABLATION || !(mat_base(m)->flags & (TRANSMISSIVE ? 16u : 32u));
    // (scalar) the btdf lobe is evaluated: not for a scalar record whose lobe constants are zero (tr_dmat::flags bit 6)
    const bool lobe_on = transmits && (!std::is_same<MatP, cdmat*>::value || TR_ABLATION || !(mat_base(m)->flags & 64u));
    auto issue_taps = [&]() {
    if (transmits) {
        const auto mb = mat_base(m);
        // refract(-v, n, ior) :248-256 ; unit length by construction (Snell), so no re-normalise
        float eta = mb->eta;
        float k = fmaf(m_neg_eta2(*mb), fmaf(-nov_raw, nov_raw, 1.0f), 1.0f);
        float cn = fmaf(-eta, nov_raw, fast_sqrt(k));   // eta * n.i + sqrt(k), n.i = -n.v
        len = mb->thickness * ns.w;                      // thickness * model_scale :264
        const float len_exit = TR_ABLATE(L, 256u) ? 0.0f : len;   // (profiling: the taps at the pixel's own place, a streaming pattern)
        float ex = fmaf(fmaf(-eta, v.x, -cn * n.x), len_exit, pos.x);
        float ey = fmaf(fmaf(-eta, v.y, -cn * n.y), len_exit, pos.y);
        float ez = fmaf(fmaf(-eta, v.z, -cn * n.z), len_exit, pos.z);
        const TR_CONSTANT float* P = L->fp.proj_view;   // column-major
        float cx = fmaf(P[8], ez, fmaf(P[4], ey, fmaf(P[0], ex, P[12])));
        float cy = fmaf(P[9], ez, fmaf(P[5], ey, fmaf(P[1], ex, P[13])));
        float cw = fmaf(P[11], ez, fmaf(P[7], ey, fmaf(P[3], ex, P[15])));
        float hw = 0.5f * rcp(cw);                      // (clip.xy / clip.w + 1) / 2  :330-332
        float tu = fmaf(cx, hw, 0.5f);
        float tv = fmaf(cy, hw, 0.5f);
        if (TR_ABLATE(L, 512u)) tu = tv = 0.5f;   // (profiling: every tap the same texels: no tap traffic, the same instructions)
        // lod = log2(framebuffer width) * roughness * clamp(2 ior - 2, 0, 1) (:334-335): the material's alone when it
        // has no texture slots (its tap record), per lane otherwise
        if constexpr (SCALAR_MATERIAL) {
            if (!TR_ABLATE(L, 1u)) pyramid_issue_record(pf, L->pyramid, as_constant(L->dtaps) + mat_index, tu, tv, L->fp.tap_row_lo, L->fp.tap_row_hi, L->tap_excess);
        } else {
            float lod = L->fp.log2_fb_width * m_rough_ior(*mb);
            if (!TR_ABLATE(L, 1u)) pyramid_issue(pf, L->pyramid, as_constant(L->levels), L->fp.pyr_levels, tu, tv, lod, wave_lane(), L->fp.tap_row_lo, L->fp.tap_row_hi, L->tap_excess);
        }
        if (TR_ABLATE(L, 1u)) { pf.r0[0] = pf.r0[1] = pf.r1[0] = pf.r1[1] = uint4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; pf.wx = pf.wy = splat(tu); pf.t = tv; pf.narrow0 = pf.narrow1 = false; }
        if constexpr (SCALAR_MATERIAL) {
            lut_line_issue(lf, L->lut_lines, L->fp.lut_wf, mb->lut_line, nov_raw);
        } else {   // per-pixel roughness: the row pair is found here, not carried through the light loop
            uint32_t row0, row1;
            lut_rows(mb->rough, L->fp.lut_height, L->fp.lut_stride, lf.fy, row0, row1);
            lut_issue(lf, L->lut_pairs, L->fp.lut_wf, row0, row1, nov_raw);
        }
    }
    };

    // ================= phases 2+3: the sun, then the clustered punctual lights =================
    // The sun writes the accumulators instead of adding to zeros (eval_light<FIRST>).
    light_acc acc;
    if (TR_ABLATION) acc = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    auto lights_phase = [&]() {
        claunch* L2 = launder(L);
        const auto m2 = mat_base(launder(m));
        {
            const float nov2 = nov * nov, onov2 = 1.0f - nov2;
            px.g_nov = v2f{fast_sqrt(fmaf(m2->a2[0], onov2, nov2)), TRANSMISSIVE ? fast_sqrt(fmaf(m2->a2[1], onov2, nov2)) : 0.0f};
        }
        // sun (lighting.rs:37-53 / 171-177)
        if (!TR_ABLATE(L2, 4u))
            eval_light<TRANSMISSIVE, !TR_ABLATION>(acc, *m2, px, {L2->fp.sun_dir[0], L2->fp.sun_dir[1], L2->fp.sun_dir[2]},
                                     {L2->fp.sun_intensity[0], L2->fp.sun_intensity[1], L2->fp.sun_intensity[2]}, lobe_on, diffuse_on);
        tile_phase<1>();
    };
    // punctual lights (lighting.rs:55-92 / 179-217): count, list and lights all through the scalar unit.  One trip of
    // the outer loop per distinct list of the tile: a single one when cl.uniform (the usual case), otherwise a waterfall
    // over the clusters of this material's lanes, each trip with exec = the lanes of that cluster.
    auto punctual = [&]() {
        claunch* L2 = launder(L);
        const auto m2 = mat_base(launder(m));
        cdlight* lights = as_constant(L2->lights);
        uint64_t pending = cl.uniform ? 0ull : ballot(true);
        do {
            uint32_t ck = cl.s_cluster, n = cl.s_num;
            if (!cl.uniform) {
                ck = (uint32_t)__builtin_amdgcn_readlane((int)cl.key, __ffsll((unsigned long long)pending) - 1);
                pending &= ~ballot(cl.key == ck);
                n = cluster_count(L2, opaque(ck));
            }
            if (cl.uniform || cl.key == ck) {
                const TR_CONSTANT uint32_t* list = cluster_entries(L2, opaque(ck));
#pragma clang loop unroll(disable)   // (left alone the first two trips are peeled: three copies of the light evaluation)
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t idx = (cl.uniform && i == 0u) ? cl.s_l0 : (cl.uniform && i == 1u) ? cl.s_l1 : list[i];
                    eval_punctual<TRANSMISSIVE>(acc, *m2, lights[idx], pos, px, lobe_on, diffuse_on);
                }
                if constexpr (!TRANSMISSIVE) lights_walked = n;
            }
        } while (pending != 0ull);
    };
    // ================= phase 4: the refraction taps, their resolve, the composite =================
    // Order: lights first, then the refraction taps and their resolve.  Issuing the taps before the light loop
    // would hide their latency inside the wave, but holds 16 + 6 vector registers across the loop; without them
    // the kernel fits 64 VGPRs = 8 waves per SIMD, and the other seven waves hide the latency better (measured:
    // 129 -> 122 us on the 4K frame, profiles/r01).
    // The light sums are resolved into the pixel's value FIRST (fifteen accumulators become three before the taps'
    // sixteen registers are requested), and the two sides of `transmits` meet again only at the emission: with one `if`
    // around the taps and another around their resolve the compiler's wait bookkeeping must assume the taps in flight on
    // the path that never issued them.
    auto tail = [&]() -> f3 {
        tile_phase<2>();
        TR_PROBE_DRAIN
        TR_PROBE_SINCE(t_taps)
        claunch* L4 = launder(L);
        MatP mo = launder(m);                  // what a base-colour texture changes is read through `mo`,
        const auto m4 = mat_base(mo);          // every other constant from the record
        f3 out;
        if constexpr (std::is_same<MatP, cdmat*>::value) {
            // the material's composite constants with transmission_factor folded in (tr_dmat::kd ...)
            // (the opaque pass, `fragment`, knows no transmission: plain c_diff)
            const TR_CONSTANT float* kd = TRANSMISSIVE ? m4->kd : m4->c_diff;
            out = {fmaf(m4->ks_df[0], acc.sp.x, m4->ks_f0[0] * acc.s.x), fmaf(m4->ks_df[1], acc.sp.y, m4->ks_f0[1] * acc.s.y),
                   fmaf(m4->ks_df[2], acc.sp.z, m4->ks_f0[2] * acc.s.z)};            // specular: sum I nol D*V F
            if (diffuse_on) out = {fmaf(kd[0], acc.d.x, out.x), fmaf(kd[1], acc.d.y, out.y), fmaf(kd[2], acc.d.z, out.z)};
            if (transmits) {
                if (lobe_on) {
                    out.x = fmaf(m4->kta[0], acc.ta.x, out.x);
                    out.y = fmaf(m4->kta[1], acc.ta.y, out.y);
                    out.z = fmaf(m4->kta[2], acc.ta.z, out.z);
                    out.x = fmaf(-m4->ktb[0], acc.tb.x, out.x);
                    out.y = fmaf(-m4->ktb[1], acc.tb.y, out.y);
                    out.z = fmaf(-m4->ktb[2], acc.tb.z, out.z);
                }
                issue_taps();
                TR_PROBE_WAITED(2, t_taps)
                tile_phase<3>();
                // ---- ibl_volume_refraction, part 2 (:337-353)
                f3 T = pyramid_resolve(pf);
                if (m4->flags & 1u) {  // apply_volume_attenuation (Beer's law) :275-290
                    T.x *= fast_exp2(m4->neg_atten_log2[0] * len);
                    T.y *= fast_exp2(m4->neg_atten_log2[1] * len);
                    T.z *= fast_exp2(m4->neg_atten_log2[2] * len);
                }
                const v2f AB = lut_line_resolve(lf);
                const float w = fmaf(-m4->f90, AB.y, 1.0f);      // 1 - (f0 A + f90 B), channel by channel
                out.x = fmaf(m4->kt[0], fmaf(-m4->f0[0], AB.x, w) * T.x, out.x);
                out.y = fmaf(m4->kt[1], fmaf(-m4->f0[1], AB.x, w) * T.y, out.y);
                out.z = fmaf(m4->kt[2], fmaf(-m4->f0[2], AB.x, w) * T.z, out.z);
            } else {
                TR_PROBE_WAITED(2, t_taps)
                tile_phase<3>();
            }
            out = {out.x + m4->emission[0], out.y + m4->emission[1], out.z + m4->emission[2]};
        } else {
            if constexpr (SCALAR_MATERIAL)   // (the lite class: its lights ran against the scalar record, see eval_light)
                acc.s = {fmaf(m4->ks_df[0], acc.sp.x, m4->ks_f0[0] * acc.s.x), fmaf(m4->ks_df[1], acc.sp.y, m4->ks_f0[1] * acc.s.y),
                         fmaf(m4->ks_df[2], acc.sp.z, m4->ks_f0[2] * acc.s.z)};
            f3 diffuse = {acc.d.x * mat_c_diff(mo, 0), acc.d.y * mat_c_diff(mo, 1), acc.d.z * mat_c_diff(mo, 2)};
            if (transmits) {
                issue_taps();
                TR_PROBE_WAITED(2, t_taps)
                tile_phase<3>();
                // ---- ibl_volume_refraction, part 2 (:337-353)
                f3 T = pyramid_resolve(pf);
                if (m4->flags & 1u) {  // apply_volume_attenuation (Beer's law) :275-290
                    T.x *= fast_exp2(m4->neg_atten_log2[0] * len);
                    T.y *= fast_exp2(m4->neg_atten_log2[1] * len);
                    T.z *= fast_exp2(m4->neg_atten_log2[2] * len);
                }
                v2f AB;
                if constexpr (SCALAR_MATERIAL) AB = lut_line_resolve(lf);
                else AB = lut_resolve(lf, lf.fy);
                // (1 - (f0*A + f90*B)) * attenuated * base_colour, summed with the btdf lobes
                const float fb = m4->f90 * AB.y;
                const float bx = fmaf(-mat_bt_b(m4, 0), acc.tb.x, mat_bt_a(m4, 0) * acc.ta.x);   // sum over lights of transmission_btdf
                const float by = fmaf(-mat_bt_b(m4, 1), acc.tb.y, mat_bt_a(m4, 1) * acc.ta.y);
                const float bz = fmaf(-mat_bt_b(m4, 2), acc.tb.z, mat_bt_a(m4, 2) * acc.ta.z);
                float tx = fmaf(1.0f - fmaf(m4->f0[0], AB.x, fb), T.x, bx) * mat_diffuse(mo, 0);
                float ty = fmaf(1.0f - fmaf(m4->f0[1], AB.x, fb), T.y, by) * mat_diffuse(mo, 1);
                float tz = fmaf(1.0f - fmaf(m4->f0[2], AB.x, fb), T.z, bz) * mat_diffuse(mo, 2);
                // lib.rs:157-159: real = tf * transmission; diffuse = lerp(diffuse, real, tf)
                float tf = m4->transmission_factor;
                diffuse.x = fmaf(fmaf(tf, tx, -diffuse.x), tf, diffuse.x);
                diffuse.y = fmaf(fmaf(tf, ty, -diffuse.y), tf, diffuse.y);
                diffuse.z = fmaf(fmaf(tf, tz, -diffuse.z), tf, diffuse.z);
            } else {
                TR_PROBE_WAITED(2, t_taps)
                tile_phase<3>();
            }
            out = {diffuse.x + acc.s.x + m4->emission[0], diffuse.y + acc.s.y + m4->emission[1],
                   diffuse.z + acc.s.z + m4->emission[2]};
        }
        if constexpr (!TRANSMISSIVE) {
            if (L4->fp.debug_clusters != 0u) {  // lib.rs:241-245
                f3 a = debug_colour_for_id(lights_walked), b = debug_colour_for_id(cl.cluster);
                out = {fmaf(b.x - 0.5f, 0.025f, a.x), fmaf(b.y - 0.5f, 0.025f, a.y), fmaf(b.z - 0.5f, 0.025f, a.z)};
            }
        }
        return out;
    };
    TR_PROBE_SINCE(t_lights)
    lights_phase();   // the sun
    punctual();
    TR_PROBE_WAITED(4, t_lights)
    after_lights();   // (a per-lane record: what only the tail reads comes back from LDS, see shade_pixel_textured)
    return tail();
}

// ------------------------------------------------------------------------ one pixel of a textured material
// Differences inside the pixel's 2x2 quad of the two values the reference's shaders differentiate
// (OpDPdx / OpDPdy): -view_vector (shader/src/lighting.rs:237) and uv (the implicit LOD of every texture fetch).
struct quad_derivs {
    uv_derivs uv;   // (full-class kernels: these and the differences of -view_vector wait in LDS, see kParkSlots)
};
// Full-class kernels: what a pixel's sampling front end reads of its tile — the quad differences, the interpolated normal
// and scale, uv — is parked in the wave's LDS ONCE per tile, for all 64 lanes, before the wave splits by material: a lane's
// column of slots is its own, so the lanes of the second material of a straddling tile find their inputs untouched by the
// first material's pixels, and nothing of it is held in registers across a pixel (20 registers; reading the planes again
// for the second material instead measured 3.5 % slower).  A lane's pixel then reuses its column for what it samples.
enum : uint32_t {
    kParkDp = 0u,        // 0-5   d(-view)/dx, d(-view)/dy           | 0-13 the sampled factors once the pixel is under way
    kParkNormal = 6u,    // 6-9   interpolated normal, model scale
    kParkUv = 10u,       // 10-11 uv
    kParkDuv = 12u,      // 12-15 du/dx, dv/dx, du/dy, dv/dy        | 14-16 the mapped normal
    kParkMapped = 14u,
};

// The front end of `fragment` / `fragment_transmission` for a material with texture slots (lib.rs:65-76, 120-124,
// 190-194; lighting.rs:222-313): the (scalar) material record says which slots are bound, the taps of every bound
// texture are issued together, then the sampled factors are digested per lane and the pixel continues through
// the same shade_pixel as an untextured one.
template <bool TRANSMISSIVE, uint32_t SLOTS /* the slots a material of this launch may bind: kSlotsAll / kSlotsMid */>
__device__ __forceinline__ f3 shade_pixel_textured(claunch* L, uint32_t material, cdmat* dm, float4 pd, uint32_t lane,
                                                   const cluster_list& cl_in, const float* __restrict__ lds_srgb,
                                                   float* lds_park TR_PROBE_ARGS_DECL) {
    L = launder(L);
    dm = launder(dm);
    // What the sampling front end produces slot by slot waits in the wave's LDS (one float per lane and value) instead of
    // in registers, and what only the end of the pixel reads stays there across the light loop (after_lights below).
    float* const park = lds_park + lane;
    auto put = [&](uint32_t f, float v) { park[f * 64u] = v; };
    auto get = [&](uint32_t f) { return park[f * 64u]; };
    constexpr uint32_t kNx = kParkMapped, kNy = kParkMapped + 1u, kNz = kParkMapped + 2u;
    // the pixel's inputs, from the slots the kernel parked them in for the whole tile
    float4 ns = float4{get(kParkNormal), get(kParkNormal + 1u), get(kParkNormal + 2u), get(kParkNormal + 3u)};
    const float2 uv = float2{get(kParkUv), get(kParkUv + 1u)};
    quad_derivs qd;
    qd.uv = {get(kParkDuv), get(kParkDuv + 1u), get(kParkDuv + 2u), get(kParkDuv + 3u)};
    const TR_CONSTANT tr_material_info* mi = as_constant(L->materials) + material;
    cdtex* tex = as_constant(L->textures);
    const uint32_t* __restrict__ arena = L->tex_arena;
    constexpr auto may = [](int k) { return ((SLOTS >> k) & 1u) != 0u; };
    const int32_t id_diffuse = may(0) ? mi->textures.diffuse : -1, id_mr = may(1) ? mi->textures.metallic_roughness : -1;
    const int32_t id_normal = may(2) ? mi->textures.normal_map : -1, id_emissive = may(3) ? mi->textures.emissive : -1;
    const int32_t id_transmission = TRANSMISSIVE && may(4) ? mi->textures.transmission : -1;
    const int32_t id_thickness = TRANSMISSIVE && may(5) ? mi->textures.thickness : -1;
    const int32_t id_specular = may(6) ? mi->textures.specular : -1, id_spec_colour = may(7) ? mi->textures.specular_colour : -1;
    // (what a slot that cannot be bound would modulate is never parked: it stays the record's scalar value)

    // Sampling geometry (LOD, level pair, wrapped tap coordinates, weights) depends on the texture's size only, and a
    // material's textures usually share one size: it is computed for the first bound slot and kept while the following
    // slots have that size (they only issue their eight taps: scalar chain base + the shared per-lane offsets); a slot of
    // another size recomputes it in place.  All the conditions are scalar (ids and sizes come from the material record).
    const int32_t ids[8] = {id_diffuse, id_mr, id_normal, id_emissive, id_transmission, id_thickness, id_specular, id_spec_colour};
    tex_geom g0;
    uint32_t w0 = 0u, h0 = 0u;   // (scalar) the size g0 was computed for
    // ONE slot at a time: its eight taps are issued, filtered and dropped before the next slot's are requested (a
    // compiler barrier between slots), and what a slot yields is materialised on the spot.  With the slots' taps in
    // flight together — or their filters sunk down to the late uses of emission, thickness, the transmission factor —
    // the kernel needed 122 registers = 4 waves per SIMD; the memory-level parallelism given up inside a wave comes back
    // as resident waves.
    // use(S): S(channel) filters one channel of the slot's taps (call only when the slot is bound)
    auto with_slot = [&](auto slot, auto&& use) {
        constexpr int k = decltype(slot)::value;
        cdtex* t = tex + ids[k];
        const bool srgb = t->srgb != 0u;
        if (t->width != w0 || t->height != h0) {
            w0 = t->width;
            h0 = t->height;
            tex_geom_compute(g0, t, uv.x, uv.y, qd.uv);
        }
        tex_taps taps;
        texture_issue_shared(taps, arena, t, g0);
        use([&](auto ch) { return texture_resolve_shared<decltype(ch)::value>(taps, g0, srgb, lds_srgb); });
    };
    auto slot_done = []() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> c0;
    typedef std::integral_constant<int, 1> c1;
    typedef std::integral_constant<int, 2> c2;
    typedef std::integral_constant<int, 3> c3;

    // calculate_normal + compute_cotangent_frame (lighting.rs:222-259) FIRST: the differences of the view vector and the
    // interpolated normal end here, and the mapped normal waits in LDS while the other slots are sampled
    if (id_normal != -1) {
        const float inv_n = rsq(dot3(ns.x, ns.y, ns.z, ns.x, ns.y, ns.z));
        const f3 n = {ns.x * inv_n, ns.y * inv_n, ns.z * inv_n};
        // map_normal * 255/127 - 128/127 (the compiled shader folds 255/127 into one multiply)
        float mx = 0.0f, my = 0.0f, mz = 0.0f;
        with_slot(std::integral_constant<int, 2>{}, [&](auto S) {
            mx = fmaf(S(c0{}), 255.0f / 127.0f, -128.0f / 127.0f);
            my = fmaf(S(c1{}), 255.0f / 127.0f, -128.0f / 127.0f);
            mz = fmaf(S(c2{}), 255.0f / 127.0f, -128.0f / 127.0f);
        });
        asm volatile("" : "+v"(mx), "+v"(my), "+v"(mz));
        slot_done();
        auto cross = [](f3 a, f3 b) { return f3{fmaf(a.y, b.z, -(b.y * a.z)), fmaf(a.z, b.x, -(b.z * a.x)), fmaf(a.x, b.y, -(b.x * a.y))}; };
        const f3 dp_dx = {get(kParkDp), get(kParkDp + 1u), get(kParkDp + 2u)}, dp_dy = {get(kParkDp + 3u), get(kParkDp + 4u), get(kParkDp + 5u)};
        const f3 dp2perp = cross(dp_dy, n), dp1perp = cross(n, dp_dx);
        const f3 t = {fmaf(dp2perp.x, qd.uv.dudx, dp1perp.x * qd.uv.dudy), fmaf(dp2perp.y, qd.uv.dudx, dp1perp.y * qd.uv.dudy),
                      fmaf(dp2perp.z, qd.uv.dudx, dp1perp.z * qd.uv.dudy)};
        const f3 b = {fmaf(dp2perp.x, qd.uv.dvdx, dp1perp.x * qd.uv.dvdy), fmaf(dp2perp.y, qd.uv.dvdx, dp1perp.y * qd.uv.dvdy),
                      fmaf(dp2perp.z, qd.uv.dvdx, dp1perp.z * qd.uv.dvdy)};
        const float invmax = rsq(fmaxf(dot3(t.x, t.y, t.z, t.x, t.y, t.z), dot3(b.x, b.y, b.z, b.x, b.y, b.z)));
        const float tx = mx * invmax, by_ = my * invmax;            // Mat3::from_cols(t * invmax, b * invmax, n) * map_normal
        ns.x = fmaf(n.x, mz, fmaf(b.x, by_, t.x * tx));
        ns.y = fmaf(n.y, mz, fmaf(b.y, by_, t.y * tx));
        ns.z = fmaf(n.z, mz, fmaf(b.z, by_, t.z * tx));             // shade_pixel normalises
        asm volatile("" : "+v"(ns.x), "+v"(ns.y), "+v"(ns.z));
    }
    put(kNx, ns.x); put(kNy, ns.y); put(kNz, ns.z);
    slot_done();
    lane_dmat lm;
    // diffuse = diffuse_factor * sample (lib.rs:65-69)
    float dr = mi->diffuse_factor[0], dg = mi->diffuse_factor[1], db = mi->diffuse_factor[2];
    if (id_diffuse != -1) {
        with_slot(std::integral_constant<int, 0>{}, [&](auto S) { dr *= S(c0{}); dg *= S(c1{}); db *= S(c2{}); });
        asm volatile("" : "+v"(dr), "+v"(dg), "+v"(db));
        slot_done();
    }
    put(0, dr); put(1, dg); put(2, db);
    slot_done();
    // get_material_params (lighting.rs:261-301)
    float metallic = mi->metallic_factor, rough = mi->roughness_factor;
    if (id_mr != -1) {
        with_slot(std::integral_constant<int, 1>{}, [&](auto S) { metallic *= S(c2{}); rough *= S(c1{}); });   // "These two are switched!"
        asm volatile("" : "+v"(metallic), "+v"(rough));
        slot_done();
    }
    put(3, metallic); put(4, rough);
    slot_done();
    float scx = mi->specular_colour_factor[0], scy = mi->specular_colour_factor[1], scz = mi->specular_colour_factor[2];
    if (id_spec_colour != -1) {
        with_slot(std::integral_constant<int, 7>{}, [&](auto S) { scx *= S(c0{}); scy *= S(c1{}); scz *= S(c2{}); });
        asm volatile("" : "+v"(scx), "+v"(scy), "+v"(scz));
        slot_done();
    }
    if constexpr (may(7)) { put(5, scx); put(6, scy); put(7, scz); }
    slot_done();
    float specular_factor = mi->specular_factor;
    if (id_specular != -1) {
        with_slot(std::integral_constant<int, 6>{}, [&](auto S) { specular_factor *= S(c3{}); });
        asm volatile("" : "+v"(specular_factor));
        slot_done();
    }
    if constexpr (may(6)) put(8, specular_factor);
    slot_done();
    // get_emission (lighting.rs:303-313)
    lm.emission[0] = mi->emissive_factor[0];
    lm.emission[1] = mi->emissive_factor[1];
    lm.emission[2] = mi->emissive_factor[2];
    if (id_emissive != -1) {
        with_slot(std::integral_constant<int, 3>{}, [&](auto S) { lm.emission[0] *= S(c0{}); lm.emission[1] *= S(c1{}); lm.emission[2] *= S(c2{}); });
        asm volatile("" : "+v"(lm.emission[0]), "+v"(lm.emission[1]), "+v"(lm.emission[2]));
        slot_done();
    }
    if constexpr (may(3)) { put(9, lm.emission[0]); put(10, lm.emission[1]); put(11, lm.emission[2]); }
    slot_done();
    lm.transmission_factor = mi->transmission_factor;               // lib.rs:71-77
    if (id_transmission != -1) {
        with_slot(std::integral_constant<int, 4>{}, [&](auto S) { lm.transmission_factor *= S(c0{}); });
        asm volatile("" : "+v"(lm.transmission_factor));
        slot_done();
    }
    if constexpr (may(4)) put(12, lm.transmission_factor);
    slot_done();
    lm.thickness = mi->thickness_factor;                            // lib.rs:120-124
    if (id_thickness != -1) {
        with_slot(std::integral_constant<int, 5>{}, [&](auto S) { lm.thickness *= S(c1{}); });
        asm volatile("" : "+v"(lm.thickness));
        slot_done();
    }
    if constexpr (may(5)) put(13, lm.thickness);
    slot_done();
    lm.eta = dm->eta;
    lm.neg_atten_log2[0] = dm->neg_atten_log2[0];
    lm.neg_atten_log2[1] = dm->neg_atten_log2[1];
    lm.neg_atten_log2[2] = dm->neg_atten_log2[2];

    // The per-lane record is digested only now, when every slot has been sampled and the sampling geometry, the quad
    // differences and the taps are dead: the digest's twenty values and the sampling state are never live together.
    slot_done();
    dr = get(0); dg = get(1); db = get(2);
    metallic = get(3); rough = get(4);
    if constexpr (may(7)) { scx = get(5); scy = get(6); scz = get(7); }
    if constexpr (may(6)) specular_factor = get(8);
    if constexpr (may(4)) lm.transmission_factor = get(12);
    lm.flags = (dm->flags & 1u) | (lm.transmission_factor != 0.0f ? 2u : 0u);
    digest_factors<false>(lm, metallic, rough, dm->ior_clamp, dm->f0_dielectric, specular_factor, scx, scy, scz, dr, dg, db,
                          L->fp.lut_height, L->fp.lut_stride);
    lm.metallic = metallic;
    lm.rough = rough;
    const float4 ns2 = float4{get(kNx), get(kNy), get(kNz), ns.w};
    // What only the end of the pixel reads — base colour, metallic, roughness, emission, transmission factor, thickness —
    // stays in its LDS slot across the light loop and is read back behind it: ten registers the loop does not hold.
    auto after_lights = [&]() {
        // (through a laundered pointer: these are new loads to the optimiser, neither merged with the digest's reads of
        //  the same slots nor a barrier to the scalar loads the tail wants early)
        const float* again = park;
        asm volatile("" : "+v"(again));
        auto reget = [&](uint32_t f) { return again[f * 64u]; };
        lm.diffuse[0] = reget(0); lm.diffuse[1] = reget(1); lm.diffuse[2] = reget(2);
        lm.metallic = reget(3); lm.rough = reget(4);
        if constexpr (may(3)) { lm.emission[0] = reget(9); lm.emission[1] = reget(10); lm.emission[2] = reget(11); }
        if constexpr (may(4)) lm.transmission_factor = reget(12);
        if constexpr (may(5)) lm.thickness = reget(13);
    };
    return shade_pixel<TRANSMISSIVE, const lane_dmat*>(L, &lm, material, pd, ns2, lane, cl_in TR_PROBE_ARGS, after_lights);
}

// ------------------------------------------------------------------------ one pixel of a "lite" textured material
// The front end for a material of the lite class (lite_dmat): one implicit-LOD tap set of the base-colour texture
// (lib.rs:65-69, 190-194), then the untextured pixel with the sampled colour.
template <bool TRANSMISSIVE>
__device__ __forceinline__ f3 shade_pixel_lite(claunch* L, uint32_t material, cdmat* dm, float4 pd, float4 ns, float2 uv,
                                               const uv_derivs& duv, uint32_t lane, const cluster_list& cl,
                                               const float* __restrict__ lds_srgb TR_PROBE_ARGS_DECL) {
    L = launder(L);
    const TR_CONSTANT tr_material_info* mi = as_constant(L->materials) + material;
    cdtex* t = as_constant(L->textures) + mi->textures.diffuse;
    tex_geom g;
    tex_taps taps;
    TR_PROBE_SINCE(t_tex)
    tex_geom_compute(g, t, uv.x, uv.y, duv);
    const bool srgb = t->srgb != 0u;
    lite_dmat lm;
    lm.m = dm;
    if (ballot(g.frac != 0.0f) == 0ull) {   // (uniform) every pixel of the wave exactly on its lower level
        texture_issue_shared<1>(taps, L->tex_arena, t, g);
        lm.diffuse[0] = mi->diffuse_factor[0] * texture_resolve_shared<0, 1>(taps, g, srgb, lds_srgb);
        lm.diffuse[1] = mi->diffuse_factor[1] * texture_resolve_shared<1, 1>(taps, g, srgb, lds_srgb);
        lm.diffuse[2] = mi->diffuse_factor[2] * texture_resolve_shared<2, 1>(taps, g, srgb, lds_srgb);
    } else {
        texture_issue_shared(taps, L->tex_arena, t, g);
        lm.diffuse[0] = mi->diffuse_factor[0] * texture_resolve_shared<0>(taps, g, srgb, lds_srgb);
        lm.diffuse[1] = mi->diffuse_factor[1] * texture_resolve_shared<1>(taps, g, srgb, lds_srgb);
        lm.diffuse[2] = mi->diffuse_factor[2] * texture_resolve_shared<2>(taps, g, srgb, lds_srgb);
    }
    // The colour is first USED at the end of the pixel; left to itself the optimiser sinks the whole filter down there
    // and keeps the eight taps, their weights and the decode look-ups alive across the light loop (+40 registers).
    asm volatile("" : "+v"(lm.diffuse[0]), "+v"(lm.diffuse[1]), "+v"(lm.diffuse[2]));
    TR_PROBE_WAITED(3, t_tex)
    return shade_pixel<TRANSMISSIVE, const lite_dmat*>(L, &lm, material, pd, ns, lane, cl TR_PROBE_ARGS);
}

// ---- tonemap ----------------------------------------------------------------------
// fragment_tonemap (shader/src/lib.rs:683-697) + LottesTonemapper::tonemap (shader/src/tonemapping.rs:8-27);
// the fullscreen triangle samples texel centres, so the input is the HDR texel itself.  pow = exp2(y log2 x).
__device__ __forceinline__ float fast_pow(float x, float y) { return fast_exp2(y * fast_log2(x)); }

// The sRGB target's encode (fixed function) of exp2(l), for l = log2 of a value already clamped to [0, 1] (l <= 0, -inf
// for zero): one v_exp_f32 for either branch of the transfer function — 12.92 x below the knee, 1.055 x^(1/2.4) - 0.055 above.
__device__ __forceinline__ uint32_t log2_to_srgb8(float l) {
    const bool low = l <= -8.3192688f;                       // log2(0.0031308)
    const float e = fast_exp2(low ? l : l * (1.0f / 2.4f));
    const float v = low ? 12.92f * e : fmaf(1.055f, e, -0.055f);
    return (uint32_t)fmaf(v, 255.0f, 0.5f);
}

// fragment_tonemap (shader/src/lib.rs:683-697, shader/src/tonemapping.rs:8-27) + the sRGB encode of the swapchain
// format.  Two pixels per thread (16-byte loads, 8-byte stores).  The operator is a chain of powers: it is evaluated in
// the log2 domain, so that consecutive powers share their logarithm — peak^a and (peak^a)^d both from log2(peak);
// (ratio^cs * peak').clamp(0, 1) goes straight into the sRGB curve's own power as cs log2(ratio) + log2(peak') — 19
// transcendental instructions per pixel instead of 26 (the kernel is bound by them), and fewer roundings.  Black pixels
// (0 / 0 in the reference: NaN, which its min(1).max(0) turns into 1) come out the same way: log2(0) - log2(0) is NaN
// and v_min_f32(NaN, 0) is 0.
__device__ __forceinline__ uint32_t tonemap_pixel(uint32_t lo, uint32_t hi, const tr_tonemap_params& p, float e1, int bgra) {
    const float r = h2f_lo(lo), g = h2f_hi(lo), b = h2f_lo(hi);
    const float mx = fmaxf(r, fmaxf(g, b));
    const float lmx = fast_log2(mx);
    const float la = p.a * lmx;
    const float z = fast_exp2(la);                                  // peak^a
    const float tm = z * rcp(fmaf(fast_exp2(la * p.d), p.b, p.c));  // tonemap_inner: z / (z^d b + c)
    const float ltm = fast_log2(tm);
    const float t = fast_exp2(ltm * p.crosstalk);                   // tonemapped_max^crosstalk
    const float c[3] = {r, g, b};
    uint32_t out8[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float x = fast_exp2((fast_log2(c[k]) - lmx) * e1);          // (color / max)^(saturation / cross_saturation)
        x = fmaf(1.0f - x, t, x);                                   // lerp(ratio, 1, tonemapped_max^crosstalk)
        const float l = fmaf(fast_log2(x), p.cross_saturation, ltm);   // log2(ratio^cross_saturation * tonemapped_max)
        out8[k] = log2_to_srgb8(fminf(l, 0.0f));                    // .min(ONE); .max(ZERO) is exp2's own range
    }
    return bgra ? (out8[2] | (out8[1] << 8) | (out8[0] << 16) | 0xFF000000u) : (out8[0] | (out8[1] << 8) | (out8[2] << 16) | 0xFF000000u);
}

// ------------------------------------------------------------------------ the shading kernel
// Grid: 8 * k workgroups, k per XCD (hardware workgroup b runs on XCD b % 8), of one wave each, kGridRounds times what
// is resident.  The 64x4-pixel block tiles of the rect are cut into 8
// contiguous bands, one per XCD; the wave in slot w of its XCD takes the 16x4 quarters w, w + W, ... of the band, so an
// XCD sweeps its band front to back and its L2 serves a compact window of the screen (and of the opaque pyramid behind
// it).  A wave is a 16x4 pixel tile — few waves straddle a material or cluster border, every plane row segment is
// still >= one 128 B line.
//
// TEXTURED (chosen by the host when an uploaded material has texture slots): additionally reads the uv plane,
// forms the quad differences with two lane swizzles (the 16x4 wave tile holds whole 2x2 quads: partner lanes are
// lane^1 and lane^16; the host guarantees an even rect origin) before the wave splits by material, and sends
// materials flagged as textured through shade_pixel_textured.  The sRGB decode table sits in LDS.
// Which planes a plane launch loads non-temporally (bit 0 the position plane, 1 the normal plane, 2 the ids, 3 the uv plane; see
// fetch): textured launches the normal + uv planes (all four: +3 %), the transmissive pass both float4 planes, the opaque pass
// position + ids.
constexpr uint32_t planes_nt_mask(bool textured, bool transmissive) { return textured ? 10u : transmissive ? 3u : 5u; }
template <class T, bool NT>
__device__ __forceinline__ T ld_plane(const void* base, uint32_t byte_offset) {
    if constexpr (NT) return ld_stream<T>(base, byte_offset);
    else return ld<T>(base, byte_offset);
}
constexpr uint32_t kWaveTileW = 16u, kWaveTileH = 4u;                       // wave tile: 16x4 pixels
constexpr uint32_t kBlockTileW = 4u * kWaveTileW, kBlockTileH = kWaveTileH;   // block tile: four of them side by side
constexpr uint32_t kGridRounds = 4u;   // waves in the grid per resident wave (static launches)
struct tile_regs {
    float4 pd, ns;
    float2 uv;                                        // TEXTURED only
    uint32_t mat, cluster_x, cluster_y_term, px, py;  // (the two table values are only added when used)
    uint32_t cover;                                   // (scalar) the block tile's coverage word
    uint32_t cover_front;                             // (scalar) VIS opaque: the transmissive layer's word of the tile
};

// Occupancy targets for the register allocator.  The untextured transmissive variant fed from visibility words fits 64
// registers (8 waves) once told that eight waves are wanted.  The full-class variants hold 66-73 registers since their
// tile inputs and late-read factors wait in LDS (kParkDp ...; round 3: 96-100 = 5 waves); LDS (17 slots + the sRGB table
// = 5376 B per wave, allocated in 1280-byte granules: 22 or 26 slots measured alike, 8 % slower) then admits 25 waves per
// CU, and the hint keeps the allocator from spending registers it has no use for.  Every other variant is left to
// itself: forced up, the textured classes spill, and scratch costs more than the waves give (DESIGN.md 3.1;
// tests/test_kernel_resources.py holds the line).
#define TR_WAVES_ATTR __attribute__((amdgpu_waves_per_eu((TEX == kTexNone && TRANSMISSIVE && VIS) ? 8 : (TEX >= kTexFull && TRANSMISSIVE) ? 6 : 1)))
// TEX: which material classes the uploaded materials hold (the host knows: tr_upload_materials) — ONE launch shades them all:
//   0  no material has a texture slot: every material through the scalar record;
//   1  untextured materials and the LITE class (lite_dmat: only a base-colour texture, dielectric);
//   2  every class, the FULL class among them (any other combination of texture slots: shade_pixel_textured).  Since the
//      full-class pixel holds no more registers than the lite one (its tile inputs and late-read factors wait in LDS, see
//      kParkDp), a material set that mixes classes no longer takes a launch per class (rounds 2-3: a TEX = 1 launch, a tile
//      list, a full-class launch whose latency a handful of tiles could not fill — 4K mesh frame 222 -> 215 us, demo glTF
//      229 -> 214);
//   3  as 2, for material sets whose full-class materials bind nothing but the base-colour, the metallic-roughness and the
//      normal-map slot (the usual glTF set): what the other five slots would modulate stays the material record's scalar
//      value instead of a per-lane one.
constexpr int kTexNone = 0, kTexLite = 1, kTexAll = 2, kTexAllMid = 3;
constexpr int kTexFull = kTexAll;   // (>= kTexFull: the launch carries the full class's sampling front end)
constexpr uint32_t kSlotsAll = 0xFFu, kSlotsMid = 0x07u;   // bit k: slot k of shade_pixel_textured's `ids` may be bound

// VIS (the frame recorder's launches): a pixel's inputs are interpolated here from the rasteriser's visibility word and
// triangle record (vis_interpolate, the resolve's own arithmetic) instead of being read from TGB-v1 planes, which the
// frame then never writes: 60 B per covered pixel less traffic (8 B word read + zeroed instead of 8 + 8 + 44 written by
// the resolve and 44 read back here).  The launch zeroes the word of every pixel it shades for the next frame.
template <bool TRANSMISSIVE, typename OutT /* uint2 = RGBA16F, float4 = RGBA32F */, int TEX = kTexNone, bool VIS = false>
__global__ __launch_bounds__(64) TR_WAVES_ATTR void shade_kernel(const tr_launch launch_by_value) {
    constexpr bool TEXTURED = TEX != kTexNone;
    constexpr uint32_t kPlanesNt = planes_nt_mask(TEXTURED, TRANSMISSIVE);
    (void)launch_by_value;  // read through the kernarg segment pointer, see tr_launch
    claunch* L = launder((claunch*)__builtin_amdgcn_kernarg_segment_ptr());
    __shared__ float lds_srgb[TEXTURED ? 256 : 1];
    __shared__ float lds_park[TEX >= kTexFull ? kParkedValues * 64u : 64u];
    const uint32_t lane = threadIdx.x;   // one wave per workgroup
    if constexpr (TEXTURED) {
        // The sRGB decode table goes global -> LDS without passing through registers (global_load_lds, 4 x 64 dwords) and
        // is not waited for here: the wave's first wait is for its first tile's inputs, requested behind these transfers,
        // and vector memory returns in order — a wave's start costs one round trip less (a frame is ~24 000 waves of ~5 tiles).
        const float* table = L->srgb_to_linear;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + k * 64u + lane),
                                             (__attribute__((address_space(3))) void*)(lds_srgb + k * 64u), 4, 0, 0);
    }
    const uint32_t lx = lane & (kWaveTileW - 1u), ly = lane / kWaveTileW;   // position inside the wave's tile
    // (full-class kernels, one wave per workgroup: the lane index is derived where it is used — wave_lane — instead of
    //  held across the pixel, and so is everything that depends on it alone)
    auto lane_here = [&]() -> uint32_t {
        if constexpr (TEX >= kTexFull) return wave_lane();
        else return lane;
    };

    // One contiguous band of the rect's block tiles (row-major) per XCD.  Dealing the tiles to the XCDs in k smaller
    // chunks or in stripes of tile rows, to average the scene's cost variations over the XCDs (the most loaded XCD runs
    // ~3 % longer than the mean on the 4K frame), measured equal or slower for every k: each L2 then serves k windows
    // of the screen and of the pyramid behind it.
    const uint32_t ntiles = L->fp.tiles_x * L->fp.tiles_y;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t per = ntiles >> 3, rem = ntiles & 7u;
    // VIS launches (real frames, where whole screen regions are empty or cheap): the XCDs are dealt STRIPES of
    // kStripeTileRows tile rows in turn instead of one contiguous band each, so that every XCD gets its share of the
    // covered part of the screen (a frame whose upper half is sky left half of the XCDs idle).
    const bool listed = VIS && TRANSMISSIVE && L->front_list != nullptr;   // (scalar) walk the opaque launch's list of covered block tiles
    const uint32_t listed_tiles = listed ? as_constant(L->front_list_count)[blockIdx.x & (kFrontLists - 1u)] : 0u;
    const bool striped = (VIS || TEX != kTexNone) && !listed;
    uint32_t stripes_own = 0u, striped_len = 0u;
    if (striped) {
        const uint32_t st = L->fp.stripe_tiles;
        const uint32_t nstripes = (ntiles + st - 1u) / st;
        stripes_own = xcd < nstripes ? (nstripes - 1u - xcd) / 8u + 1u : 0u;
        const bool owns_last = stripes_own != 0u && ((nstripes - 1u) & 7u) == xcd;
        striped_len = stripes_own * st - (owns_last ? nstripes * st - ntiles : 0u);
    }
    const uint32_t band_start = listed ? 0u : xcd * per + min(xcd, rem);
    const uint32_t band_len = listed ? listed_tiles : striped ? striped_len : per + (xcd < rem ? 1u : 0u);

    // Out-of-rect lanes read a clamped (valid) pixel and are masked later.  There is no software prefetch of the
    // next tile: its 13 registers cost two of the eight resident waves per SIMD, and even scheduled so that nothing
    // younger is waited for during the light evaluation it measured slower (121 vs 113 us).
    // The tile's per-pixel inputs (position, normal, uv, material id, cluster table entries) of the pixels t.px, t.py.
    auto load_inputs = [&](claunch* F, tile_regs& t, uint32_t tile) {
        const uint32_t cx = min(t.px, F->fp.rect_x1 - 1u), cy = min(t.py, F->fp.rect_y1 - 1u);
        const uint32_t gpix = mad24(cy - F->fp.g_origin_y, F->fp.g_width, cx - F->fp.g_origin_x);
        if (TR_ABLATE(F, 64u)) {  // profiling only: no G-buffer traffic (synthetic per-lane inputs)
            t.mat = (tile >> 5) & 15u;
            t.pd = float4{(float)cx * 1e-3f - 1.5f, 2.0f + (float)cy * 1e-3f, -2.0f, 0.004f};
            t.ns = float4{0.1f, 0.3f + (float)lx * 1e-2f, 0.9f, 1.0f};
            t.cluster_x = 5; t.cluster_y_term = 0;
            return;
        }
        // The planes are read once: the two float4 planes are loaded non-temporally, so that the stream does not displace
        // the pyramid texels, LUT lines and cluster lists the kernel keeps re-reading from L2; the ids are ordinary loads.
        // Measured with COLD inputs (bench.py: every step a different input set; 4K frame as two bands): both float4 planes
        // non-temporal 83.5 us, all three planes 83.6, position only 87.0 (rounds 2-3's choice, tuned on ONE re-read input
        // set, where a plane kept in the Infinity Cache is worth more than a clean L2: 78 vs 82 us), normal + ids 88.2,
        // none 90.4.  The opaque pass keeps position + ids (its stores are re-read): planes_nt_mask.
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef float f2v __attribute__((ext_vector_type(2)));
        if constexpr (VIS) {
            // (whole-frame launches only: the rect is the frame, its pitch the visibility buffer's)
            const unsigned long long key = ld<unsigned long long>(F->vis, gpix * 8u);
            // (the cluster table entries do not depend on the word: requested with it, ahead of the triangle's planes)
            t.cluster_x = ld<uint32_t>(F->cluster_x, cx * 4u);
            t.cluster_y_term = ld<uint32_t>(F->cluster_y_term, cy * 4u);
            t.mat = TR_NOT_COVERED;
            t.pd = t.ns = float4{0.f, 0.f, 0.f, 0.f};
            t.uv = float2{0.f, 0.f};
            if (key != 0ull) {
                vis_fragment v;
                vis_planes_interpolate(F->tri_planes[(uint32_t)key], key, cx, cy, v);
                t.pd = float4{v.position[0], v.position[1], v.position[2], v.depth};
                t.ns = float4{v.normal[0], v.normal[1], v.normal[2], v.scale};
                if constexpr (TEXTURED) t.uv = float2{v.uv[0], v.uv[1]};
                t.mat = v.material_id;
            }
            return;
        }
        t.mat = ld_plane<uint32_t, (kPlanesNt & 4u) != 0u>(F->material_id, gpix * 4u);
        { const f4v a = ld_plane<f4v, (kPlanesNt & 1u) != 0u>(F->pos_depth, gpix * 16u); t.pd = float4{a.x, a.y, a.z, a.w}; }
        { const f4v a = ld_plane<f4v, (kPlanesNt & 2u) != 0u>(F->nrm_scale, gpix * 16u); t.ns = float4{a.x, a.y, a.z, a.w}; }
        if constexpr (TEXTURED) { const f2v a = ld_plane<f2v, (kPlanesNt & 8u) != 0u>(F->uv, gpix * 8u); t.uv = float2{a.x, a.y}; }
        t.cluster_x = ld<uint32_t>(F->cluster_x, cx * 4u);
        t.cluster_y_term = ld<uint32_t>(F->cluster_y_term, cy * 4u);
    };
    // (scalar) wave tile j of this XCD's band -> block tile j / 4 (64x4 pixels) of the rect, and the wave tile's column / row
    auto tile_of = [&](claunch* F, uint32_t j, uint32_t& tile, uint32_t& txi, uint32_t& tyi) {
        tile = listed ? as_constant(F->front_list)[(blockIdx.x & (kFrontLists - 1u)) * F->front_list_cap + (j >> 2)] : band_start + (j >> 2);
        if (striped) {   // local tile q of this XCD: stripe q / stripe_tiles of its own, i.e. stripe (that * 8 + xcd) of the frame
            const uint32_t q = j >> 2, st = F->fp.stripe_tiles;
            uint32_t own = __umulhi(q, F->fp.stripe_magic), off = q - own * st;
            if (off >= st) {
                off -= st;
                ++own;
            }
            tile = (own * 8u + xcd) * st + off;
        }
        // tile / tiles_x without the vector unit: q = mulhi(tile, floor(2^32 / d)) is the quotient or one less
        tyi = __umulhi(tile, F->fp.tiles_x_magic);
        txi = tile - tyi * F->fp.tiles_x;
        if (txi >= F->fp.tiles_x) {
            txi -= F->fp.tiles_x;
            ++tyi;
        }
        if (!VIS && F->fp.strip_tile_rows != 0u) {   // (scalar) this rank's strips of a frame shared with other ranks (never a VIS launch)
            const uint32_t T = F->fp.strip_tile_rows;
            uint32_t k = __umulhi(tyi, F->fp.strip_magic), r = tyi - k * T;
            if (r >= T) {
                r -= T;
                ++k;
            }
            tyi = (k * F->fp.strip_world + F->fp.strip_rank) * T + r;
        }
        txi = txi * 4u + (j & 3u);
    };
    // j = wave tile of this XCD's band: block tile j / 4 (64x4 pixels), quarter j % 4
    auto fetch = [&](uint32_t j, tile_regs& t) {
        claunch* F = launder(L);
        uint32_t tile, txi, tyi;
        tile_of(F, j, tile, txi, tyi);
        if constexpr (TEX >= kTexFull) {
            // (the lane's place in the tile is derived again per tile: two instructions instead of two registers held —
            //  or spilled — across the whole pixel)
            const uint32_t l = lane_here();
            t.px = F->fp.rect_x0 + txi * kWaveTileW + (l & (kWaveTileW - 1u));
            t.py = F->fp.rect_y0 + tyi * kWaveTileH + l / kWaveTileW;
        } else {
            t.px = F->fp.rect_x0 + txi * kWaveTileW + lx;
            t.py = F->fp.rect_y0 + tyi * kWaveTileH + ly;
        }
        // (scalar) the coverage word of the block tile, when the frame recorder rasterised the layer itself: 0 = nothing
        // landed there (the tile is skipped without touching its inputs); bit 1 = fragments of a full-class material
        // (their tile inputs are parked in LDS, see kParkDp), bit 2 = of any other (raster_kernel)
        uint32_t cover;
        if constexpr (VIS && !TRANSMISSIVE) {
            // (a VIS launch always has its layer's map, and the opaque one the transmissive layer's: the pointers come in one
            //  scalar round trip, pinned together, the two words in the next — no null checks in the chain)
            const uint32_t* own_words = F->tile_cover;
            const uint32_t* front_words = F->cover_front;
            uint32_t* list = F->front_list_build;
            cover = as_constant(own_words)[tile];
            t.cover_front = as_constant(front_words)[tile];
            // (once per block tile: by the wave of its first quarter)
            if (list && t.cover_front != 0u && (j & 3u) == 0u && lane_here() == 0u) {
                const uint32_t sub = tile & (kFrontLists - 1u);   // (at most ceil(tiles / kFrontLists) entries: front_list_cap)
                list[sub * F->front_list_cap + atomicAdd(F->front_list_build_count + sub, 1u)] = tile;
            }
        } else if constexpr (VIS) {
            cover = as_constant(F->tile_cover)[tile];
        } else {
            cover = F->tile_cover ? as_constant(F->tile_cover)[tile] : 0xFFFFFFFFu;
        }
        t.cover = cover;
        if (cover == 0u) {
            t.mat = TR_NOT_COVERED;
            t.pd = t.ns = float4{0.f, 0.f, 0.f, 0.f};
            t.uv = float2{0.f, 0.f};
            t.cluster_x = t.cluster_y_term = 0u;
            return;
        }
        load_inputs(F, t, tile);
    };

    TR_PROBE_WAVE_BEGIN
    // Which tile next: static — the wave in slot w of its XCD takes the 16x4 tiles w, w + W, w + 2W, ... of the band
    // (W = waves of the XCD in the grid; four neighbouring waves cover one 64x4 block tile side by side, so their plane
    // rows are 1 KB contiguous and their stores 512 B).  Handing tiles out dynamically balances the waves (static: the
    // longest-lived wave of the 4K frame runs 36 % longer than the mean) but measured slower.
    const uint32_t wave_tiles = band_len * 4u;
    const uint32_t slot = listed ? blockIdx.x / kFrontLists : (blockIdx.x >> 3);
    uint32_t j = slot;
    tile_regs cur;
    uint32_t clear_present = 0u;         // (scalar) the presented background pixel, once the wave has met a background tile
    bool have_clear_present = false;
    while (j < wave_tiles) {
        tile_phase<0>();
        TR_PROBE_SINCE(t_fetch)   // (from before the fetch: a VIS tile's inputs are two dependent round trips inside it)
        fetch(j, cur);
        TR_PROBE_WAITED(0, t_fetch)
        TR_PROBE_TILE_DONE
        claunch* S = launder(L);
        const bool inside = cur.px < S->fp.rect_x1 && cur.py < S->fp.rect_y1;
        const bool active = inside && cur.mat != TR_NOT_COVERED;
        const uint32_t key = inside ? cur.mat : TR_NOT_COVERED;   // the material of a lane that has work
        f3 out = {0.f, 0.f, 0.f};  // clear colour of the opaque pass (src/main.rs:1592-1601)
        uint64_t todo = ballot(key != TR_NOT_COVERED);
        cdmat* dmats = as_constant(S->dmats);
        if (TR_ABLATE(S, 32u)) {  // profiling only: pure streaming skeleton
            todo = 0;
            out = f3{cur.pd.x + cur.ns.x + (float)cur.mat, cur.pd.y + cur.ns.y + (float)cur.cluster_x, cur.pd.z + cur.ns.z + cur.pd.w + cur.ns.w};
        } else if (todo) {
            // the light lists of all 64 pixels are requested at once, before the wave splits by material
            TR_PROBE_SINCE(t_cluster)
            const cluster_list cl = cluster_lookup(S, cur.pd.w, cur.cluster_x + cur.cluster_y_term, key != TR_NOT_COVERED);
            TR_PROBE_WAITED(1, t_cluster)
            // One material at a time through the scalar unit; a wave that straddles k materials loops k times.
            quad_derivs qd;
            if constexpr (TEXTURED) {
                // dFdx = value(x|1) - value(x&~1), dFdy likewise, zero where the partner has no fragment (lanes
                // clamped to the frame edge read the same pixel as their partner: zero as well)
                const bool covered = cur.mat != TR_NOT_COVERED;
                auto swz_x = [](float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x041F)); };  // lane ^ 1
                auto swz_y = [](float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); };  // lane ^ 16: the pixel below / above
                // (every swizzle is evaluated by the whole wave, outside any condition: a lane-crossing read under a
                // short-circuit would run with the uncovered lanes switched off and read zeros from them)
                const int mat_x = __builtin_amdgcn_ds_swizzle((int)cur.mat, 0x041F);
                const int mat_y = __builtin_amdgcn_ds_swizzle((int)cur.mat, 0x401F);
                const bool cov_x = covered & (mat_x != (int)TR_NOT_COVERED), cov_y = covered & (mat_y != (int)TR_NOT_COVERED);
                const uint32_t lq = lane_here();   // (derived per tile, not carried: see fetch)
                const float sgn_x = (lq & 1u) ? -1.0f : 1.0f, sgn_y = (lq & kWaveTileW) ? -1.0f : 1.0f;
                const float nvx = -(S->fp.view_position[0] - cur.pd.x), nvy = -(S->fp.view_position[1] - cur.pd.y),
                            nvz = -(S->fp.view_position[2] - cur.pd.z);
                auto ddx = [&](float v) { const float d = (swz_x(v) - v) * sgn_x; return cov_x ? d : 0.0f; };
                auto ddy = [&](float v) { const float d = (swz_y(v) - v) * sgn_y; return cov_y ? d : 0.0f; };
                qd.uv = {ddx(cur.uv.x), ddx(cur.uv.y), ddy(cur.uv.x), ddy(cur.uv.y)};
                if (TEX >= kTexFull && (cur.cover & 2u)) {   // (scalar: the tile holds full-class fragments; see kParkDp)
                    float* const park = lds_park + lq;
                    auto put = [&](uint32_t f, float v) { park[f * 64u] = v; };
                    put(kParkDp, ddx(nvx)); put(kParkDp + 1u, ddx(nvy)); put(kParkDp + 2u, ddx(nvz));
                    put(kParkDp + 3u, ddy(nvx)); put(kParkDp + 4u, ddy(nvy)); put(kParkDp + 5u, ddy(nvz));
                    put(kParkNormal, cur.ns.x); put(kParkNormal + 1u, cur.ns.y); put(kParkNormal + 2u, cur.ns.z); put(kParkNormal + 3u, cur.ns.w);
                    put(kParkUv, cur.uv.x); put(kParkUv + 1u, cur.uv.y);
                    put(kParkDuv, qd.uv.dudx); put(kParkDuv + 1u, qd.uv.dvdx); put(kParkDuv + 2u, qd.uv.dudy); put(kParkDuv + 3u, qd.uv.dvdy);
                    asm volatile("" ::: "memory");
                }
            }
            while (todo) {
                const int l0 = __ffsll((unsigned long long)todo) - 1;
                const uint32_t mk = (uint32_t)__builtin_amdgcn_readlane((int)key, l0);
                const uint32_t m0 = opaque(mk);   // tables are indexed with m0, the branch compares mk (see opaque())
                const uint64_t group = ballot(key == mk);
                todo &= ~group;
                if (key == mk) {
                    if constexpr (TEXTURED) {
                        const uint32_t fl = dmats[m0].flags;   // (scalar) 4: full class, 8: lite, neither: no texture slot
                        if (TEX >= kTexFull && (fl & 12u) == 4u)
                            out = shade_pixel_textured<TRANSMISSIVE, TEX == kTexAllMid ? kSlotsMid : kSlotsAll>(
                                L, m0, dmats + m0, cur.pd, lane_here(), cl, lds_srgb, lds_park TR_PROBE_ARGS);
                        else if (fl & 8u)
                            out = shade_pixel_lite<TRANSMISSIVE>(L, m0, dmats + m0, cur.pd, cur.ns, cur.uv, qd.uv, lane_here(), cl, lds_srgb TR_PROBE_ARGS);
                        else
                            out = shade_pixel<TRANSMISSIVE>(L, dmats + m0, m0, cur.pd, cur.ns, lane_here(), cl TR_PROBE_ARGS);
                    } else {
                        out = shade_pixel<TRANSMISSIVE>(L, dmats + m0, m0, cur.pd, cur.ns, lane, cl TR_PROBE_ARGS);
                    }
                }
            }
        } else {
            // (a tile without work has its table loads arrive here: left in flight on this one path, they make the compiler
            //  wait for EVERYTHING outstanding — the previous tile's store with them — before the next tile's first load)
            if constexpr (TEX == kTexNone && !VIS) asm volatile("" :: "v"(cur.cluster_x), "v"(cur.cluster_y_term));   // (the last two loads issued)
        }
        // (There is no prefetch of the next tile in front of this tile's store — rounds 2, 4 and 5 built it three ways, planes
        //  in registers, planes and visibility words in LDS: never faster, docs/TRIED_r05.md.  The next tile's first wait is
        //  therefore also a wait for this store's acknowledgement: vector memory completes in order.)
        const uint32_t out_px = cur.px, out_py = cur.py;
        j += launder(L)->fp.j_step;
        if constexpr (VIS && !TRANSMISSIVE && sizeof(OutT) == 8) {
            // A wave tile of the frame recorder's opaque launch that no opaque fragment and no transmissive one landed in — the
            // background: half of a typical frame — holds the clear colour, texel for texel: its targets are written from
            // constants (level 1's box of four clear texels is the clear texel), and the presented pixel — fragment_tonemap of
            // that texel: 19 transcendental instructions — is evaluated once per WAVE, not per tile (the same function on the
            // same bits), and kept in a scalar register.
            if (ballot(active) == 0ull && cur.cover_front == 0u) {
                claunch* W = launder(L);
                const uint2 clear = uint2{0x00000000u, 0x3C000000u};   // RGBA16F (0, 0, 0, 1), src/main.rs:1592-1601
                // (the presented clear texel is evaluated by the WHOLE wave, ahead of the divergent `inside`: evaluated under it,
                //  the lanes outside the rect on the wave's first background tile kept 0 and presented 0 on every later tile —
                //  frames whose width is no multiple of 16 or whose height no multiple of 4, from about 3 Mpixels up)
                if (W->present != nullptr && !have_clear_present) {
                    const tr_tonemap_params pp = {W->present_params.a, W->present_params.b, W->present_params.c, W->present_params.d,
                                                  W->present_params.crosstalk, W->present_params.saturation, W->present_params.cross_saturation};
                    clear_present = (uint32_t)__builtin_amdgcn_readfirstlane((int)tonemap_pixel(clear.x, clear.y, pp, W->present_e1, W->present_bgra));
                    have_clear_present = true;
                }
                if (inside) {
                    const uint32_t pix = mad24(out_py, W->fp.width, out_px);
                    st<uint2>(W->hdr, pix * 8u, clear);
                    if (W->mip0) st<uint2>(W->mip0, pix * 8u, clear);
                    if (W->mip1 != nullptr && (lane_here() & 17u) == 0u)
                        st<uint2>(W->mip1, mad24(out_py >> 1, W->fp.width >> 1, out_px >> 1) * 8u, clear);
                    if (W->present != nullptr) st<uint32_t>(W->present, pix * 4u, clear_present);
                }
                continue;
            }
        }
        // transmissive pass: uncovered pixels keep the attachment (LOAD); opaque pass: clear colour
        const bool write = TRANSMISSIVE ? active : inside;
        bool final_colour = true;   // (VIS) no later launch of the frame writes this pixel
        if constexpr (VIS) {
            // the reader of a visibility word leaves it zeroed for the next frame (see the template's comment)
            // (the zero is formed where it is stored: hoisted, it is a register pair held — or spilled — across the whole pixel)
            unsigned long long zero = 0ull;
            asm volatile("" : "+v"(zero));
            if (active) st<unsigned long long>(launder(L)->vis, mad24(out_py, launder(L)->fp.width, out_px) * 8u, zero);
            if constexpr (!TRANSMISSIVE) {
                // (scalar) a transmissive fragment landed in this tile: a pixel this launch shades knows its opaque depth —
                // the transmissive winner stays only if it is nearer (depth GREATER, reversed Z); over a pixel without an
                // opaque fragment it always stays
                claunch* V = launder(L);
                if (cur.cover_front != 0u && (active || (V->present != nullptr && inside))) {
                    const uint32_t at = mad24(out_py, V->fp.width, out_px) * 8u;
                    const unsigned long long front = ld<unsigned long long>(V->vis_front, at);
                    if (active && front != 0ull && !(__uint_as_float((uint32_t)(front >> 32)) > cur.pd.w))
                        st<unsigned long long>(V->vis_front, at, zero);
                    else
                        final_colour = front == 0ull;
                }
            }
        }
        if constexpr (!TRANSMISSIVE && sizeof(OutT) == 8) {
            // Level 1 of the opaque pyramid straight from the values being stored: for even sizes a LINEAR blit is the
            // 2x2 box of the ROUNDED level-0 texels, (q00/4 + q10/4) + (q01/4 + q11/4) (box4 / the oracle), and a wave
            // tile holds whole quads — the mip chain then never reads level 0 (66 of its 88 MB at 4K).
            claunch* M = launder(L);
            uint2* const mip1 = M->mip1;
            if (mip1 != nullptr) {
#pragma clang fp contract(off)
                const uint2 q = pack_rgba16f(out.x, out.y, out.z, 1.0f);
                const float c[4] = {h2f_lo(q.x), h2f_hi(q.x), h2f_lo(q.y), h2f_hi(q.y)};
                float s4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float partner = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(c[k]), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]: lane ^ 1
                    const float hsum = c[k] * 0.25f + partner * 0.25f;
                    const float below = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(hsum), 0x401F));               // lane ^ 16
                    s4[k] = hsum + below;
                }
                if ((lane_here() & 17u) == 0u && inside)
                    st<uint2>(mip1, mad24(cur.py >> 1, M->fp.width >> 1, cur.px >> 1) * 8u, pack_rgba16f(s4[0], s4[1], s4[2], s4[3]));
            }
        }
        if (write && !(TR_ABLATE(S, 128u) && out.x != 12345.0f)) {  // bit7: profiling, no stores
            claunch* W = launder(L);
            const uint32_t pix = mad24(out_py, W->fp.width, out_px);
            if constexpr (sizeof(OutT) == 8) {
                const uint2 o = pack_rgba16f(out.x, out.y, out.z, 1.0f);
                if constexpr (TRANSMISSIVE) {   // the last writer of the frame: streamed out past L2 (the opaque pass's
                    typedef uint32_t u2v __attribute__((ext_vector_type(2)));   // targets are re-read at once: cached)
                    __builtin_nontemporal_store(u2v{o.x, o.y}, reinterpret_cast<u2v*>(static_cast<char*>(W->hdr) + pix * 8u));
                } else if (final_colour) {
                    // (a pixel under a surviving transmissive fragment is written again by the frame's transmissive launch,
                    //  which does not read the target there: only level 0 of the pyramid needs its opaque colour)
                    st<uint2>(W->hdr, pix * 8u, o);
                }
                if constexpr (VIS) {
                    if (W->present != nullptr && final_colour) {
                        const tr_tonemap_params pp = {W->present_params.a, W->present_params.b, W->present_params.c, W->present_params.d,
                                                      W->present_params.crosstalk, W->present_params.saturation, W->present_params.cross_saturation};
                        st<uint32_t>(W->present, pix * 4u, tonemap_pixel(o.x, o.y, pp, W->present_e1, W->present_bgra));
                    }
                }
            } else {
                st<OutT>(W->hdr, pix * 16u, OutT{out.x, out.y, out.z, 1.0f});
            }
            if constexpr (!TRANSMISSIVE) {
                uint2* mip0 = W->mip0;
                if (mip0) st<uint2>(mip0, pix * 8u, pack_rgba16f(out.x, out.y, out.z, 1.0f));
            }
        }
    }
    TR_PROBE_WAVE_END
}

// LightClusterCoefficients::get_depth_slice over an array (tr_get_depth_slice): the passes' own device function.
__global__ __launch_bounds__(256) void depth_slice_kernel(const float* __restrict__ depth, uint32_t count, const slice_params sp,
                                                          uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const float d = depth[min(i, count - 1u)];   // (every lane of the wave runs depth_slice: it ballots)
    const uint32_t z = depth_slice(sp, d);
    if (i < count) out[i] = z;
}

// ------------------------------------------------------------------------ tap records
// One thread per material: tr_dtap from the digested material, the pyramid's level table and log2(framebuffer width) —
// the operations shade_pixel used to run per tile (contraction off), so the level pair and the weight are what they were.
__global__ void digest_taps_kernel(const tr_dmat* __restrict__ dmats, const tr_level_table* __restrict__ lv, uint32_t levels,
                                   float log2_fb_width, tr_dtap* __restrict__ out, uint32_t count) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float lod = log2_fb_width * dmats[i].rough_ior;               // glam-pbr/src/lib.rs:334-335
    const float l = fminf(fmaxf(lod, 0.0f), (float)(levels - 1u));      // sampler: lod in [0, levels - 1], NaN -> 0
    const float lf = floorf(l);
    tr_dtap d;
    d.t = l - lf;
    const uint32_t pair[2] = {(uint32_t)lf, min((uint32_t)lf + 1u, levels - 1u)};
    d.narrow = 0u;
    for (int k = 0; k < 2; ++k) {
        const uint32_t w = lv->width[pair[k]], h = lv->height[pair[k]];
        d.level[k] = pair[k];
        d.offset[k] = lv->offset[pair[k]] * 8u;
        d.pitch[k] = h >= 2u ? w * 8u : 0u;
        d.width[k] = w;
        d.wf[k] = (float)w;
        d.hf[k] = (float)h;
        d.xhi[k] = (float)(w - 1u);
        d.yhi[k] = (float)(h - 1u);
        d.xlim[k] = w >= 2u ? (float)(w - 2u) : 0.0f;
        d.ylim[k] = h >= 2u ? (float)(h - 2u) : 0.0f;
        if (w < 2u) d.narrow |= 1u << k;
    }
    d._pad[0] = d._pad[1] = 0u;
    out[i] = d;
}

// ------------------------------------------------------------------------ material digestion
// One thread per material; runs once per tr_upload_materials.  Same fp32 operations as the
// reference where a value is a pure function of MaterialInfo (glam-pbr/src/lib.rs:141-161,
// 182-198, 425-435; shader/src/lighting.rs:261-313 with every texture id == -1).
__global__ void digest_materials_kernel(const tr_material_info* __restrict__ in, tr_dmat* __restrict__ out,
                                        uint32_t count, uint32_t lut_height, uint32_t lut_stride) {
#pragma clang fp contract(off)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const tr_material_info mi = in[i];
    tr_dmat d;
    const float ior = mi.index_of_refraction;
    const float root = (ior - 1.0f) / (ior + 1.0f);
    d.f0_dielectric = root * root;                                   // to_dielectric_f0 :190-195
    d.ior_clamp = fminf(fmaxf(ior * 2.0f - 2.0f, 0.0f), 1.0f);
    digest_factors<true>(d, mi.metallic_factor, mi.roughness_factor, d.ior_clamp, d.f0_dielectric, mi.specular_factor,
                   mi.specular_colour_factor[0], mi.specular_colour_factor[1], mi.specular_colour_factor[2],
                   mi.diffuse_factor[0], mi.diffuse_factor[1], mi.diffuse_factor[2], lut_height, lut_stride);
    for (int k = 0; k < 3; ++k) d.emission[k] = mi.emissive_factor[k];
    d.eta = 1.0f / ior;
    d.neg_eta2 = -(d.eta * d.eta);
    d.transmission_factor = mi.transmission_factor;
    d.thickness = mi.thickness_factor;
    const bool has_atten = !(mi.attenuation_distance == __builtin_inff());
    const tr_textures& t = mi.textures;
    const bool textured = t.diffuse != -1 || t.metallic_roughness != -1 || t.normal_map != -1 || t.emissive != -1 ||
                          t.transmission != -1 || t.thickness != -1 || t.specular != -1 || t.specular_colour != -1;
    // the lite class (bit 3, always with bit 2): only the base-colour slot is bound and the material is a dielectric
    const bool lite = t.diffuse != -1 && t.metallic_roughness == -1 && t.normal_map == -1 && t.emissive == -1 &&
                      t.transmission == -1 && t.thickness == -1 && t.specular == -1 && t.specular_colour == -1 &&
                      mi.metallic_factor == 0.0f;
    d.flags = (has_atten ? 1u : 0u) | (mi.transmission_factor != 0.0f ? 2u : 0u) | (textured ? 4u : 0u) | (lite ? 8u : 0u);
    {   // bits 4, 5: the diffuse constant of the pass is +-0 in every channel (never for NaN factors: those must propagate)
        const float omtf = 1.0f - mi.transmission_factor;
        bool kd0 = true, cd0 = true;
        for (int k = 0; k < 3; ++k) {
            kd0 = kd0 && d.c_diff[k] * omtf == 0.0f;
            cd0 = cd0 && d.c_diff[k] == 0.0f;
        }
        bool bt0 = true;   // (never for NaN / infinite factors either: 0 * x must stay what it is)
        for (int k = 0; k < 3; ++k) bt0 = bt0 && d.bt_a[k] == 0.0f && d.bt_b[k] == 0.0f;
        d.flags |= (kd0 ? 16u : 0u) | (cd0 ? 32u : 0u) | (bt0 ? 64u : 0u);
    }
    for (int k = 0; k < 3; ++k) {
        float coeff = -logf(mi.attenuation_colour[k]) / mi.attenuation_distance;  // :284
        d.neg_atten_log2[k] = has_atten ? (-coeff) * kLog2e : 0.0f;
    }
    d.lut_line = i * lut_stride;
    d._pad = 0u;
    d._pad3 = d._pad4 = d._pad5 = d._pad6 = 0.0f;
    const float f0_max = fmaxf(d.f0[0], fmaxf(d.f0[1], d.f0[2]));
    d.omf0_max = 1.0f - f0_max;
    d.ndf_min = -(d.f90 - f0_max);
    for (int k = 0; k < 3; ++k) {
        d.ks_f0[k] = d.k[0] * d.f0[k];
        d.ks_df[k] = d.k[0] * d.df[k];
    }
    const float tf = mi.transmission_factor, tf2 = tf * tf;
    for (int k = 0; k < 3; ++k) {
        d.kd[k] = d.c_diff[k] * (1.0f - tf);
        d.kt[k] = tf2 * d.diffuse[k];
        d.kta[k] = d.kt[k] * d.bt_a[k];
        d.ktb[k] = d.kt[k] * d.bt_b[k];
    }
    out[i] = d;
}

// Per material, the GGX LUT at the material's roughness: entry k = (A, B) of texel clamp(k-1) and of texel clamp(k)
// after the row interpolation, already divided by 255 (what lut_resolve computes per pixel for the row pair).
__global__ void build_lut_lines_kernel(const uint32_t* __restrict__ pairs, const tr_dmat* __restrict__ dmats,
                                       float4* __restrict__ lines, uint32_t stride, uint32_t count) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, mi = blockIdx.y;
    if (k >= stride || mi >= count) return;
    const tr_dmat& m = dmats[mi];
    const uint32_t p0 = pairs[m.lut_row0 + k], p1 = pairs[m.lut_row1 + k];
    auto b = [](uint32_t w, int i) { return (float)((w >> (8 * i)) & 0xFFu); };
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = fmaf(b(p1, c) - b(p0, c), m.lut_fy, b(p0, c)) * (1.0f / 255.0f);
    lines[(size_t)mi * stride + k] = float4{v[0], v[1], v[2], v[3]};
}

// GGX LUT -> pair table.  For the unclamped left tap i0 = floor(u*w - 0.5) in [-1, w], entry
// k = i0 + 1 of a row holds (R,G) of texel clamp(i0) in its low half and of texel clamp(i0 + 1)
// in its high half: both horizontal neighbours in one dword load.
__global__ void build_lut_pairs_kernel(const uint32_t* __restrict__ rgba8, uint32_t* __restrict__ pairs, uint32_t w,
                                       uint32_t h, uint32_t stride) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t row = blockIdx.y;
    if (k >= stride || row >= h) return;
    int i0 = (int)k - 1;
    int mx = (int)w - 1;
    uint32_t a = rgba8[row * w + (uint32_t)min(max(i0, 0), mx)];
    uint32_t b = rgba8[row * w + (uint32_t)min(max(i0 + 1, 0), mx)];
    pairs[row * stride + k] = (a & 0xFFFFu) | ((b & 0xFFFFu) << 16);
}

// ------------------------------------------------------------------------ mip chain
// generate_mips (src/main.rs:2046-2064): level l from level l-1 as a LINEAR blit of the whole
// image; exact fp32 arithmetic in the oracle's order (no contraction) so the chain is
// bit-identical to the CPU restatement.  One thread per destination texel.
__global__ __launch_bounds__(256) void downsample_kernel(const uint2* __restrict__ src, uint2* __restrict__ dst,
                                                         uint32_t ws, uint32_t hs, uint32_t wd, uint32_t hd) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t j = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (i >= wd || j >= hd) return;
    const float sx = (float)ws / (float)wd, sy = (float)hs / (float)hd;
    float x = ((float)i + 0.5f) * sx - 0.5f;
    float y = ((float)j + 0.5f) * sy - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float ax = x - fx0, by = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    int x1 = min(x0 + 1, (int)ws - 1), y1 = min(y0 + 1, (int)hs - 1);
    x0 = min(max(x0, 0), (int)ws - 1);
    y0 = min(max(y0, 0), (int)hs - 1);
    const float w00 = (1.0f - ax) * (1.0f - by), w10 = ax * (1.0f - by), w01 = (1.0f - ax) * by, w11 = ax * by;
    const uint2 q00 = src[(size_t)y0 * ws + x0], q10 = src[(size_t)y0 * ws + x1];
    const uint2 q01 = src[(size_t)y1 * ws + x0], q11 = src[(size_t)y1 * ws + x1];
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto ch = [k](uint2 q) {
            uint32_t wv = (k < 2) ? q.x : q.y;
            return __half2float(__ushort_as_half((unsigned short)((k & 1) ? (wv >> 16) : (wv & 0xFFFFu))));
        };
        o[k] = (ch(q00) * w00 + ch(q10) * w10) + (ch(q01) * w01 + ch(q11) * w11);
    }
    dst[(size_t)j * wd + i] = pack_rgba16f(o[0], o[1], o[2], o[3]);
}

// ---- fused mip chain -------------------------------------------------------------------------------------
// generate_mips in two launches instead of one per level (11 at 4K, each launch bound at ~4 us):
//   * mip_even_kernel: while both sizes of the parent level are even, a LINEAR blit is an exact 2x2 box filter
//     and a 32x32 block of level 0 determines a 16x16 / 8x8 / 4x4 / 2x2 / 1x1 block of levels 1..5 on its own.
//     One workgroup reads its 32x32 tile once (16-byte loads, two texels each), keeps every level it produces
//     in LDS *as the rounded RGBA16F value the next blit would read back*, and writes each level once.
//   * mip_tail_kernel: the remaining small levels (120x67 ... 1x1 at 4K) in ONE workgroup, previous level in
//     LDS, general LINEAR weights (odd sizes), one __syncthreads per level.
// Arithmetic and rounding are those of downsample_kernel / the oracle (same products, same association).
__device__ __forceinline__ uint2 box4(uint2 q00, uint2 q10, uint2 q01, uint2 q11) {
#pragma clang fp contract(off)
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto ch = [k](uint2 q) {
            uint32_t wv = (k < 2) ? q.x : q.y;
            return __half2float(__ushort_as_half((unsigned short)((k & 1) ? (wv >> 16) : (wv & 0xFFFFu))));
        };
        o[k] = (ch(q00) * 0.25f + ch(q10) * 0.25f) + (ch(q01) * 0.25f + ch(q11) * 0.25f);
    }
    return pack_rgba16f(o[0], o[1], o[2], o[3]);
}

struct tr_mip_even_params {
    uint32_t w0, h0;        // size of the source level
    uint32_t nlevels;       // 1..5 levels to produce
    uint32_t src_offset;    // texel offsets from the pyramid base
    uint32_t dst_offset[5];
    uint32_t row_begin, row_end;   // rows of the FIRST produced level to make ([0, h0 / 2) for the whole level; a row band of a
                                   // sharded frame otherwise: row_begin a multiple of 2^(nlevels - 1), workgroups tile from it)
};

__global__ __launch_bounds__(256) void mip_even_kernel(uint2* __restrict__ pyr, const tr_mip_even_params p) {
    __shared__ uint2 lds[16 * 16 + 8 * 8 + 4 * 4 + 2 * 2 + 1];
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const uint32_t w1 = p.w0 >> 1, h1 = p.h0 >> 1;
    const uint32_t i1 = blockIdx.x * 16u + tx, j1 = p.row_begin + blockIdx.y * 16u + ty;
    uint2 r = {0u, 0u};
    if (i1 < w1 && j1 < min(h1, p.row_end)) {
        const uint2* src = pyr + p.src_offset;
        const uint4 a = *reinterpret_cast<const uint4*>(src + (size_t)(2u * j1) * p.w0 + 2u * i1);        // 16-byte aligned
        const uint4 b = *reinterpret_cast<const uint4*>(src + (size_t)(2u * j1 + 1u) * p.w0 + 2u * i1);
        r = box4(uint2{a.x, a.y}, uint2{a.z, a.w}, uint2{b.x, b.y}, uint2{b.z, b.w});
        pyr[p.dst_offset[0] + (size_t)j1 * w1 + i1] = r;
    }
    lds[ty * 16u + tx] = r;
    uint32_t base = 0, side = 16;   // LDS offset and side of the level just produced
#pragma unroll
    for (uint32_t l = 1; l < 5u; ++l) {
        lds_barrier();               // (the level just produced is handed on through LDS; its global stores are not waited for)
        if (l >= p.nlevels) break;   // uniform
        const uint32_t half = side >> 1;
        const uint32_t wl = p.w0 >> (l + 1u), hl = p.h0 >> (l + 1u);
        if (threadIdx.x < half * half) {
            const uint32_t x = threadIdx.x % half, y = threadIdx.x / half;
            const uint32_t gi = blockIdx.x * half + x, gj = (p.row_begin >> l) + blockIdx.y * half + y;
            const uint2* s = lds + base;
            const uint2 v = box4(s[(2u * y) * side + 2u * x], s[(2u * y) * side + 2u * x + 1u],
                                 s[(2u * y + 1u) * side + 2u * x], s[(2u * y + 1u) * side + 2u * x + 1u]);
            if (gi < wl && gj < min(hl, (p.row_end + (1u << l) - 1u) >> l)) pyr[p.dst_offset[l] + (size_t)gj * wl + gi] = v;
            lds[base + side * side + y * half + x] = v;
        }
        base += side * side;
        side = half;
    }
}

struct tr_mip_tail_params {
    uint32_t first, levels;                  // produce levels first .. levels-1 (first >= 1)
    uint32_t offset[TR_MAX_MIP_LEVELS], width[TR_MAX_MIP_LEVELS], height[TR_MAX_MIP_LEVELS];
};
// (a level of more than 4 096 texels is a launch of its own — downsample_kernel over as many workgroups as it takes: as the
//  tail's first level the 120x67 level of a 4K frame kept ONE workgroup computing for 5 us; 4K frame 204 -> 200 us)
constexpr uint32_t kMipTailMaxTexels = 4096;

// One destination texel of a general LINEAR blit: taps and weights (downsample_kernel's arithmetic).
struct blit_taps {
    uint32_t i00, i10, i01, i11;   // texel indices in the source level
    float w00, w10, w01, w11;
};
__device__ __forceinline__ blit_taps blit_taps_for(uint32_t t, uint32_t ws, uint32_t hs, uint32_t wd, float sx, float sy) {
#pragma clang fp contract(off)
    const uint32_t i = t % wd, j = t / wd;
    const float x = ((float)i + 0.5f) * sx - 0.5f, y = ((float)j + 0.5f) * sy - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float ax = x - fx0, by = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    const int x1 = min(x0 + 1, (int)ws - 1), y1 = min(y0 + 1, (int)hs - 1);
    x0 = min(max(x0, 0), (int)ws - 1);
    y0 = min(max(y0, 0), (int)hs - 1);
    blit_taps o;
    o.w00 = (1.0f - ax) * (1.0f - by);
    o.w10 = ax * (1.0f - by);
    o.w01 = (1.0f - ax) * by;
    o.w11 = ax * by;
    o.i00 = (uint32_t)y0 * ws + (uint32_t)x0;
    o.i10 = (uint32_t)y0 * ws + (uint32_t)x1;
    o.i01 = (uint32_t)y1 * ws + (uint32_t)x0;
    o.i11 = (uint32_t)y1 * ws + (uint32_t)x1;
    return o;
}
__device__ __forceinline__ uint2 blit_filter(const blit_taps& k, uint2 q00, uint2 q10, uint2 q01, uint2 q11) {
#pragma clang fp contract(off)
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        auto ch = [c](uint2 q) {
            uint32_t wv = (c < 2) ? q.x : q.y;
            return __half2float(__ushort_as_half((unsigned short)((c & 1) ? (wv >> 16) : (wv & 0xFFFFu))));
        };
        o[c] = (ch(q00) * k.w00 + ch(q10) * k.w10) + (ch(q01) * k.w01 + ch(q11) * k.w11);
    }
    return pack_rgba16f(o[0], o[1], o[2], o[3]);
}

// The first level of the tail reads its parent from global memory: every thread takes up to kTailBatch destination
// texels and has all their taps in flight together (one memory round trip for the level instead of one per texel of a
// thread: the level is 8 040 texels at 4K, eight per thread — this loop was 8 of the kernel's 11.7 us).
constexpr uint32_t kTailBatch = 4u;
__global__ __launch_bounds__(1024) void mip_tail_kernel(uint2* __restrict__ pyr, const tr_mip_tail_params p) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) uint2 tail_lds[];
    uint2* cur = tail_lds;                    // level l-1 (when it is in LDS)
    uint2* nxt = tail_lds;
    bool src_in_lds = false;
    for (uint32_t l = p.first; l < p.levels; ++l) {
        const uint32_t ws = p.width[l - 1], hs = p.height[l - 1], wd = p.width[l], hd = p.height[l];
        const uint2* gsrc = pyr + p.offset[l - 1];
        nxt = src_in_lds ? cur + (size_t)ws * hs : tail_lds;
        const float sx = (float)ws / (float)wd, sy = (float)hs / (float)hd;
        const uint32_t n = wd * hd;
        if (!src_in_lds) {
            for (uint32_t base = 0; base < n; base += blockDim.x * kTailBatch) {
                blit_taps k[kTailBatch];
                uint2 q[kTailBatch][4];
#pragma unroll
                for (uint32_t b = 0; b < kTailBatch; ++b) {
                    const uint32_t t = min(base + b * blockDim.x + threadIdx.x, n - 1u);
                    k[b] = blit_taps_for(t, ws, hs, wd, sx, sy);
                    q[b][0] = gsrc[k[b].i00];
                    q[b][1] = gsrc[k[b].i10];
                    q[b][2] = gsrc[k[b].i01];
                    q[b][3] = gsrc[k[b].i11];
                }
#pragma unroll
                for (uint32_t b = 0; b < kTailBatch; ++b) {
                    const uint32_t t = base + b * blockDim.x + threadIdx.x;
                    if (t < n) {
                        const uint2 v = blit_filter(k[b], q[b][0], q[b][1], q[b][2], q[b][3]);
                        pyr[p.offset[l] + t] = v;
                        nxt[t] = v;
                    }
                }
            }
        } else {
            for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
                const blit_taps k = blit_taps_for(t, ws, hs, wd, sx, sy);
                const uint2 v = blit_filter(k, cur[k.i00], cur[k.i10], cur[k.i01], cur[k.i11]);
                pyr[p.offset[l] + t] = v;
                nxt[t] = v;
            }
        }
        lds_barrier();   // (LDS only: the level's global stores are not waited for — seven store round trips otherwise)
        cur = nxt;
        src_in_lds = true;
    }
}

// ---- tonemap kernels (tonemap_pixel: above shade_kernel, whose frame-recorder launches present their pixels themselves)
// (e1 = saturation / cross_saturation, formed once on the host: the same IEEE division, ten instructions per thread fewer)
__global__ __launch_bounds__(256) void tonemap_kernel(const uint2* __restrict__ hdr, uint32_t* __restrict__ out,
                                                      uint32_t n, const tr_tonemap_params p, int bgra, const float e1) {
    const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 2u;
    if (i >= n) return;
    if (i + 1u < n) {
        const uint4 q = *reinterpret_cast<const uint4*>(hdr + i);          // pixels i, i + 1 (the target is 16-byte aligned)
        *reinterpret_cast<uint2*>(out + i) = uint2{tonemap_pixel(q.x, q.y, p, e1, bgra), tonemap_pixel(q.z, q.w, p, e1, bgra)};
    } else {
        const uint2 q = hdr[i];
        out[i] = tonemap_pixel(q.x, q.y, p, e1, bgra);
    }
}

// Four pixels per thread -> 12 bytes (three dwords) of r g b r g b ...: the frame as a sharded rank composites it.
__global__ __launch_bounds__(256) void tonemap_rgb8_kernel(const uint2* __restrict__ hdr, uint32_t* __restrict__ out, uint32_t n,
                                                           const tr_tonemap_params p, int bgra, const float e1) {
    const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (i >= n) return;
    const uint4 a = *reinterpret_cast<const uint4*>(hdr + i), b = *reinterpret_cast<const uint4*>(hdr + i + 2u);
    const uint32_t c0 = tonemap_pixel(a.x, a.y, p, e1, bgra) & 0xFFFFFFu, c1 = tonemap_pixel(a.z, a.w, p, e1, bgra) & 0xFFFFFFu;
    const uint32_t c2 = tonemap_pixel(b.x, b.y, p, e1, bgra) & 0xFFFFFFu, c3 = tonemap_pixel(b.z, b.w, p, e1, bgra) & 0xFFFFFFu;
    uint32_t* o = out + (i / 4u) * 3u;
    o[0] = c0 | (c1 << 24);
    o[1] = (c1 >> 8) | (c2 << 16);
    o[2] = (c2 >> 16) | (c3 << 8);
}

// The frame recorder's tonemap: it knows, from the rasteriser's tile coverage words of both layers, which 64x4 block tiles
// no fragment landed in — they hold the passes' clear colour, texel for texel — and a scene's background is often half
// the frame.  One workgroup covers two adjacent tiles (128x4 pixels, two pixels per thread); when one of them is
// untouched the workgroup's first wave evaluates the operator ONCE on the clear colour's texel (the same function on the
// same bits: the same output as evaluating every pixel) and the untouched tiles' threads only store it.
struct tr_tonemap_tiles {
    const uint32_t* cover[2];   // per layer: [tiles_y][tiles_x] words, 0 = no fragment in the tile
    uint32_t width, height, tiles_x;
};
__global__ __launch_bounds__(256) void tonemap_tiles_kernel(const uint2* __restrict__ hdr, uint32_t* __restrict__ out,
                                                            const tr_tonemap_params p, int bgra, const tr_tonemap_tiles tt, const float e1) {
    __shared__ uint32_t clear_out;
    const uint32_t row = threadIdx.x >> 6, col2 = threadIdx.x & 63u;
    const uint32_t tile_y = blockIdx.y, tile_x0 = blockIdx.x * 2u;
    // (scalar) the coverage of the workgroup's two tiles
    bool sky[2];
#pragma unroll
    for (uint32_t k = 0; k < 2u; ++k) {
        const uint32_t tx = min(tile_x0 + k, tt.tiles_x - 1u);
        const uint32_t t = tile_y * tt.tiles_x + tx;
        sky[k] = as_constant(tt.cover[0])[t] == 0u && as_constant(tt.cover[1])[t] == 0u;
    }
    if (sky[0] || sky[1]) {   // (uniform)
        if (threadIdx.x < 64u) {
            const uint32_t v = tonemap_pixel(0x00000000u, 0x3C000000u, p, e1, bgra);   // RGBA16F (0, 0, 0, 1)
            if (threadIdx.x == 0u) clear_out = v;
        }
        __syncthreads();
    }
    const uint32_t px = tile_x0 * 64u + col2 * 2u, py = tile_y * 4u + row;
    if (px >= tt.width || py >= tt.height) return;
    const uint32_t i = py * tt.width + px;
    const bool mine_sky = col2 < 32u ? sky[0] : sky[1];
    if (px + 1u < tt.width && ((i & 1u) == 0u)) {   // (two pixels, 16-byte aligned: even widths)
        if (mine_sky) {
            const uint32_t c = clear_out;
            *reinterpret_cast<uint2*>(out + i) = uint2{c, c};
        } else {
            const uint4 q = *reinterpret_cast<const uint4*>(hdr + i);
            *reinterpret_cast<uint2*>(out + i) = uint2{tonemap_pixel(q.x, q.y, p, e1, bgra), tonemap_pixel(q.z, q.w, p, e1, bgra)};
        }
    } else {
        for (uint32_t k = 0; k < 2u && px + k < tt.width; ++k) {
            const uint2 q = hdr[i + k];
            out[i + k] = mine_sky ? clear_out : tonemap_pixel(q.x, q.y, p, e1, bgra);
        }
    }
}

}  // namespace tr
