// tr_kernels.h — gfx950 (CDNA4, wave64) device code of the transmission/volume PBR shading path.
//
// One thread shades one pixel; a wave is a 64x1 pixel row segment so every G-buffer plane is
// read as one contiguous 1 KiB / 512 B / 256 B burst per wave; a 256-thread workgroup is a
// 64x4 screen tile.  Workgroups are renumbered so that each XCD (own L2) owns a contiguous
// band of the screen: the data-dependent taps into the opaque pyramid then re-use texel rows
// inside one L2 instead of being fetched by all eight.
//
// The math is NOT a transliteration of the reference.  Everything that depends only on the
// material is digested once per upload into a 128-byte record (f0, f90-f0, alpha^2, Beer
// coefficients, LUT row, ...; `tr_dmat`), read through the scalar unit when a wave sees one
// material.  Per light, the halfway-vector algebra is collapsed onto two dot products
// (n.l and v.l):  |v+l|^2 = 2+2 v.l,  v.h = (1+v.l)/|v+l|,  n.h = (n.v+n.l)/|v+l|, and the
// mirrored light of transmission_btdf needs no vector at all (n.l' = -n.l, v.l' = v.l-2 n.l n.v).
// D*V is one reciprocal.  x^5 is three multiplies.  These differ from the reference's op order
// by a few ulp of fp32; the parity bar is 1e-4 per-channel RMSE on the RGBA16F target
// (tests/test_gpu_parity.py), see DESIGN.md.
//
// Reference semantics implemented here (file:line relative to the reference root):
//   fragment_transmission        shader/src/lib.rs:37-162
//   fragment                     shader/src/lib.rs:164-249
//   evaluate_lights[_transmission] shader/src/lighting.rs:13-95, 145-220
//   basic_brdf / transmission_btdf / ibl_volume_refraction   glam-pbr/src/lib.rs:377-423, 200-233, 292-354
//   cluster lookup               shader/src/lib.rs:88-98, shared-structs/src/lib.rs:54-63
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/tr_shade.h"

namespace tr {

constexpr float kEpsilon = 1.1920929e-07f;  // core::f32::EPSILON (glam-pbr/src/lib.rs:95)
constexpr float kPi = 3.14159265358979323846f;
constexpr float kFrac1Pi = 0.318309886183790671538f;
constexpr float kLog2e = 1.44269504088896340736f;

// ---------------------------------------------------------------- digested material (128 B)
struct alignas(16) tr_dmat {
    float diffuse[3];      // diffuse_factor.rgb (base colour)
    float a2;              // (roughness^2)^2                       (d_ggx / v_smith alpha^2)
    float c_diff[3];       // lerp(diffuse, 0, metallic) * (1/pi)
    float at2;             // (roughness^2 * clamp(2 ior - 2, 0, 1))^2   (transmission alpha^2)
    float f0[3];           // calculate_combined_f0
    float f90;             // calculate_combined_f90 (a splat)
    float df[3];           // f90 - f0
    float eta;             // 1 / ior
    float emission[3];
    float transmission_factor;
    float neg_atten_log2[3];  // -(-ln(colour)/distance) * log2(e); 0 when distance == +INF
    float thickness;
    float rough_ior;       // roughness * clamp(2 ior - 2, 0, 1)   (pyramid lod = log2(W) * this)
    float lut_fy;          // GGX LUT row interpolation weight   (v = perceptual roughness)
    uint32_t lut_row0;     // GGX LUT row offsets into the pair table (entries)
    uint32_t lut_row1;
    uint32_t flags;        // bit0: has finite attenuation distance
    uint32_t _pad[3];
};
static_assert(sizeof(tr_dmat) == 128, "digested material is 128 B");

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
};

// Everything a shading launch needs besides the planes; passed by value (kernarg -> SGPRs).
struct tr_frame_params {
    float proj_view[16];
    float view_position[3];
    float log2_fb_width;         // log2(framebuffer_size.x as f32)
    float sun_dir[3];
    float z_near;
    float sun_intensity[3];
    float z_far;
    float cluster_size_px[2];
    float lcc_scale, lcc_bias;
    uint32_t num_clusters_x, num_clusters_y;
    uint32_t num_clusters_total;
    uint32_t debug_clusters;
    uint32_t width, height;      // frame size (colour-target pitch)
    uint32_t g_width;            // G-buffer plane pitch
    uint32_t g_origin_x, g_origin_y;  // frame position of plane element (0,0)
    uint32_t rect_x0, rect_y0, rect_x1, rect_y1;
    uint32_t tiles_x, tiles_y;   // 64x4 tiles covering the rect
    uint32_t lut_width, lut_height, lut_stride;  // pair-table stride in entries (= lut_width + 2)
    uint32_t pyr_levels;
};

struct tr_tables {
    const tr_dmat* __restrict__ dmats;
    const tr_dlight* __restrict__ lights;
    const uint32_t* __restrict__ cluster_counts;
    const uint32_t* __restrict__ light_indices;
    const uint32_t* __restrict__ lut_pairs;      // (R,G)[x-1], (R,G)[x] per entry
    const tr_level_table* __restrict__ levels;   // device copy (divergent-material path)
};

// ------------------------------------------------------------------------ small helpers
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
__device__ __forceinline__ float pow5(float x) {
    float x2 = x * x;
    return x2 * x2 * x;
}

struct f3 {
    float x, y, z;
};

// D * V with a single reciprocal, from f = 1 - noh^2 (1 - a2) supplied by the caller.
//   d_ggx                  glam-pbr/src/lib.rs:101-109   D = a2 / (pi f^2)
//   v_smith_ggx_correlated glam-pbr/src/lib.rs:114-133   V = 0.5 / g  (0 unless g > 0)
__device__ __forceinline__ float ggx_d_times_v(float f, float nov, float nol, float a2) {
    float one_minus_a2 = 1.0f - a2;
    float gv = nol * fast_sqrt(fmaf(nov * nov, one_minus_a2, a2));
    float gl = nov * fast_sqrt(fmaf(nol * nol, one_minus_a2, a2));
    float g = gv + gl;
    float dv = (a2 * (0.5f * kFrac1Pi)) * rcp(f * f * g);
    return g > 0.0f ? dv : 0.0f;
}

// f of d_ggx, well conditioned.  The reference evaluates f = noh^2 (a2 - 1) + 1 with noh = n.h; at
// low roughness (a2 ~ 1e-6) that needs 1 - noh^2 to ~1e-9 absolute, which the straightforward fp32
// form only delivers by luck of rounding.  Here 1 - noh^2 = sin2 = |n x (v+l)|^2 / |v+l|^2 comes in
// with fp32 *relative* accuracy, and f = sin2 + a2 (1 - sin2) has no cancellation.
// noh is clamped to EPSILON by Dot::new when n.h <= 0 (:93-98): then f = 1 + eps^2 (a2 - 1) = 1.
__device__ __forceinline__ float ggx_f(float sin2, float n_dot_hv, float a2) {
    return n_dot_hv > 0.0f ? fmaf(a2, 1.0f - sin2, sin2) : 1.0f;
}

// Accumulators of one pixel over its lights.
struct light_acc {
    f3 d;  // sum I * nol * (1 - max(F))            (x c_diff/pi at the end)
    f3 s;  // sum I * nol * D*V * F
    f3 t;  // sum I * (1 - F') * D_t*V_t            (x diffuse at the end)
};

// One light against one pixel: basic_brdf (+ transmission_btdf when TRANSMISSIVE).
//   n, v unit; l unit direction to the light; I = rgb intensity reaching the pixel.
template <bool TRANSMISSIVE>
__device__ __forceinline__ void eval_light(light_acc& acc, const tr_dmat& m, f3 n, f3 v, float nov_raw, float nov,
                                           f3 l, f3 I) {
    const float nl_raw = dot3(n.x, n.y, n.z, l.x, l.y, l.z);
    const float vl = dot3(v.x, v.y, v.z, l.x, l.y, l.z);
    // |n x (v+l)|^2: shared by both lobes, because the mirrored light l' = l - 2 (n.l) n differs
    // from l by a multiple of n and n x n = 0.
    const float hx = v.x + l.x, hy = v.y + l.y, hz = v.z + l.z;
    const float cx = fmaf(n.y, hz, -(n.z * hy)), cy = fmaf(n.z, hx, -(n.x * hz)), cz = fmaf(n.x, hy, -(n.y * hx));
    const float c2 = dot3(cx, cy, cz, cx, cy, cz);

    // ---- basic_brdf (glam-pbr/src/lib.rs:377-423)
    {
        float inv_h = rsq(fmaf(2.0f, vl, 2.0f));            // 1/|v+l|   (|v| = |l| = 1)
        float voh = fmaxf((1.0f + vl) * inv_h, kEpsilon);    // Dot::new clamps to EPSILON (:93-98)
        float nol = fmaxf(nl_raw, kEpsilon);
        float p = pow5(1.0f - voh);                          // fresnel_schlick :137-139
        float Fx = fmaf(m.df[0], p, m.f0[0]);
        float Fy = fmaf(m.df[1], p, m.f0[1]);
        float Fz = fmaf(m.df[2], p, m.f0[2]);
        float f = ggx_f(c2 * (inv_h * inv_h), nov_raw + nl_raw, m.a2);
        float dv = ggx_d_times_v(f, nov, nol, m.a2);
        float wd = nol * (1.0f - fmaxf(Fx, fmaxf(Fy, Fz)));  // diffuse_brdf :356-360
        float ws = nol * dv;                                 // specular_brdf :362-375
        acc.d.x = fmaf(I.x, wd, acc.d.x);
        acc.d.y = fmaf(I.y, wd, acc.d.y);
        acc.d.z = fmaf(I.z, wd, acc.d.z);
        acc.s.x = fmaf(I.x * Fx, ws, acc.s.x);
        acc.s.y = fmaf(I.y * Fy, ws, acc.s.y);
        acc.s.z = fmaf(I.z * Fz, ws, acc.s.z);
    }
    // ---- transmission_btdf (glam-pbr/src/lib.rs:200-233): light mirrored about the surface,
    //      n.l' = -(n.l), v.l' = v.l - 2 (n.l)(n.v); no vector is ever formed.
    if constexpr (TRANSMISSIVE) {
        float vlm = fmaf(-2.0f * nl_raw, nov_raw, vl);
        float inv_h = rsq(fmaf(2.0f, vlm, 2.0f));
        float voh = fmaxf((1.0f + vlm) * inv_h, kEpsilon);
        float nolm = fmaxf(-nl_raw, kEpsilon);
        float p = pow5(1.0f - voh);
        float f = ggx_f(c2 * (inv_h * inv_h), nov_raw - nl_raw, m.at2);
        float dv = ggx_d_times_v(f, nov, nolm, m.at2);
        acc.t.x = fmaf(I.x * (1.0f - fmaf(m.df[0], p, m.f0[0])), dv, acc.t.x);
        acc.t.y = fmaf(I.y * (1.0f - fmaf(m.df[1], p, m.f0[1])), dv, acc.t.y);
        acc.t.z = fmaf(I.z * (1.0f - fmaf(m.df[2], p, m.f0[2])), dv, acc.t.z);
    }
}

// Light::spotlight_factor (shared-structs/src/lib.rs:129-138); only `fragment` applies it.
__device__ __forceinline__ float spotlight_factor(const tr_dlight& L, f3 dir_to_light) {
    float theta = -dot3(dir_to_light.x, dir_to_light.y, dir_to_light.z, L.spot_dir[0], L.spot_dir[1], L.spot_dir[2]);
    return fmaxf((theta - L.cos_outer) * L.inv_spot_epsilon, 0.0f);
}

template <bool TRANSMISSIVE>
__device__ __forceinline__ void eval_punctual(light_acc& acc, const tr_dmat& m, const tr_dlight& L, f3 pos, f3 n, f3 v,
                                              float nov_raw, float nov) {
    // light_direction_and_attenuation (glam-pbr/src/lib.rs:12-23): bare 1/d^2
    float dx = L.pos[0] - pos.x, dy = L.pos[1] - pos.y, dz = L.pos[2] - pos.z;
    float d2 = dot3(dx, dy, dz, dx, dy, dz);
    float inv_d = rsq(d2);
    f3 l = {dx * inv_d, dy * inv_d, dz * inv_d};
    float att = inv_d * inv_d;
    if constexpr (!TRANSMISSIVE) {  // shader/src/lighting.rs:201-203 (absent from the transmissive loop)
        if (L.is_spot) att *= spotlight_factor(L, l);
    }
    f3 I = {L.colour[0] * att, L.colour[1] * att, L.colour[2] * att};
    eval_light<TRANSMISSIVE>(acc, m, n, v, nov_raw, nov, l, I);
}

// ------------------------------------------------------------------ opaque pyramid taps
// clamp_sampler (src/main.rs:694-705): LINEAR min/mag, LINEAR mip, CLAMP_TO_EDGE.
struct lin_tap {
    uint32_t i0, i1;
    float w;
};
__device__ __forceinline__ lin_tap linear_tap(float coord, uint32_t dim) {
    float fdim = (float)dim;
    float x = fmaf(coord, fdim, -0.5f);
    x = fminf(fmaxf(x, -1.0f), fdim);  // finite for NaN/inf coordinates; no-op otherwise (edge clamp follows)
    float fl = floorf(x);
    lin_tap t;
    t.w = x - fl;
    int a = (int)fl;
    int mx = (int)dim - 1;
    t.i0 = (uint32_t)min(max(a, 0), mx);
    t.i1 = (uint32_t)min(max(a + 1, 0), mx);
    return t;
}

__device__ __forceinline__ f3 unpack_rgb16f(uint2 t) {
    f3 r;
    r.x = __half2float(__ushort_as_half((unsigned short)(t.x & 0xFFFFu)));
    r.y = __half2float(__ushort_as_half((unsigned short)(t.x >> 16)));
    r.z = __half2float(__ushort_as_half((unsigned short)(t.y & 0xFFFFu)));
    return r;
}

__device__ __forceinline__ f3 lerp3(f3 a, f3 b, float t) {
    return {fmaf(b.x - a.x, t, a.x), fmaf(b.y - a.y, t, a.y), fmaf(b.z - a.z, t, a.z)};
}

__device__ __forceinline__ f3 bilinear_level(const uint2* __restrict__ texels, uint32_t offset, uint32_t w, uint32_t h,
                                             float u, float v) {
    lin_tap tx = linear_tap(u, w);
    lin_tap ty = linear_tap(v, h);
    const uint2* r0 = texels + offset + ty.i0 * w;
    const uint2* r1 = texels + offset + ty.i1 * w;
    uint2 q00 = r0[tx.i0], q10 = r0[tx.i1], q01 = r1[tx.i0], q11 = r1[tx.i1];
    f3 top = lerp3(unpack_rgb16f(q00), unpack_rgb16f(q10), tx.w);
    f3 bot = lerp3(unpack_rgb16f(q01), unpack_rgb16f(q11), tx.w);
    return lerp3(top, bot, ty.w);
}

// framebuffer.sample_by_lod(clamp_sampler, uv, lod).rgb (shader/src/lib.rs:135-138)
template <bool UNIFORM>
__device__ __forceinline__ f3 sample_pyramid(const uint2* __restrict__ texels, const tr_frame_params& fp,
                                             const tr_level_table* __restrict__ lv, float u, float v, float lod) {
    float l = fminf(fmaxf(lod, 0.0f), (float)(fp.pyr_levels - 1u));
    float lf = floorf(l);
    float t = l - lf;
    uint32_t l0 = (uint32_t)lf;
    uint32_t l1 = min(l0 + 1u, fp.pyr_levels - 1u);
    if constexpr (UNIFORM) {  // lod depends on the material only: wave-uniform -> scalar loads
        l0 = __builtin_amdgcn_readfirstlane(l0);
        l1 = __builtin_amdgcn_readfirstlane(l1);
    }
    f3 a = bilinear_level(texels, lv->offset[l0], lv->width[l0], lv->height[l0], u, v);
    f3 b = bilinear_level(texels, lv->offset[l1], lv->width[l1], lv->height[l1], u, v);
    return lerp3(a, b, t);
}

// textures[ggx_lut].sample(clamp_sampler, (n.v, roughness)).xy (shader/src/lib.rs:126-133).
// The row pair and its weight depend on the material only (tr_dmat); the pair table gives both
// horizontal neighbours of a row in one dword.
__device__ __forceinline__ void sample_lut(const uint32_t* __restrict__ pairs, const tr_frame_params& fp,
                                           const tr_dmat& m, float nov_raw, float& A, float& B) {
    float fw = (float)fp.lut_width;
    float x = fmaf(nov_raw, fw, -0.5f);
    x = fminf(fmaxf(x, -1.0f), fw);
    float fl = floorf(x);
    float fx = x - fl;
    uint32_t k = (uint32_t)((int)fl + 1);
    uint32_t p0 = pairs[m.lut_row0 + k];
    uint32_t p1 = pairs[m.lut_row1 + k];
    float r00 = (float)(p0 & 0xFFu), g00 = (float)((p0 >> 8) & 0xFFu);
    float r10 = (float)((p0 >> 16) & 0xFFu), g10 = (float)(p0 >> 24);
    float r01 = (float)(p1 & 0xFFu), g01 = (float)((p1 >> 8) & 0xFFu);
    float r11 = (float)((p1 >> 16) & 0xFFu), g11 = (float)(p1 >> 24);
    float rt = fmaf(r10 - r00, fx, r00), rb = fmaf(r11 - r01, fx, r01);
    float gt = fmaf(g10 - g00, fx, g00), gb = fmaf(g11 - g01, fx, g01);
    A = fmaf(rb - rt, m.lut_fy, rt) * (1.0f / 255.0f);
    B = fmaf(gb - gt, m.lut_fy, gt) * (1.0f / 255.0f);
}

// ------------------------------------------------------------------------ cluster lookup
// shader/src/lib.rs:88-98; LightClusterCoefficients::get_depth_slice shared-structs/src/lib.rs:54-63.
// The two screen divisions and the linear-depth division are IEEE-exact so that cluster
// boundaries fall on the same pixels as in the reference.
__device__ __forceinline__ uint32_t f32_as_u32_sat(float f) {  // Rust `as u32`: saturating, NaN -> 0
    return (f > 0.0f) ? ((f >= 4294967296.0f) ? 0xFFFFFFFFu : (uint32_t)f) : 0u;
}
__device__ __forceinline__ uint32_t cluster_index(const tr_frame_params& fp, uint32_t px, uint32_t py, float depth) {
#pragma clang fp contract(off)
    uint32_t cx = f32_as_u32_sat(((float)px + 0.5f) / fp.cluster_size_px[0]);
    uint32_t cy = f32_as_u32_sat(((float)py + 0.5f) / fp.cluster_size_px[1]);
    float depth_range = 2.0f * (1.0f - depth) - 1.0f;
    float lin = (2.0f * fp.z_near * fp.z_far) / ((fp.z_far + fp.z_near) - depth_range * (fp.z_far - fp.z_near));
    uint32_t cz = f32_as_u32_sat(fmaxf(__log2f(lin) * fp.lcc_scale + fp.lcc_bias, 0.0f));
    return cz * fp.num_clusters_x * fp.num_clusters_y + cy * fp.num_clusters_x + cx;
}

__device__ __forceinline__ uint2 pack_rgba16f(float r, float g, float b, float a) {
    __half2 lo = __floats2half2_rn(r, g);
    __half2 hi = __floats2half2_rn(b, a);
    uint2 o;
    o.x = *reinterpret_cast<uint32_t*>(&lo);
    o.y = *reinterpret_cast<uint32_t*>(&hi);
    return o;
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

// ------------------------------------------------------------------------ one pixel
// UNIFORM: the whole wave shares `m` (scalar registers).  Returns rgb; alpha is 1.
template <bool TRANSMISSIVE, bool UNIFORM>
__device__ __forceinline__ f3 shade_pixel(const tr_frame_params& fp, const tr_tables& tb, const tr_dmat& m,
                                          const uint2* __restrict__ pyramid, float4 pd, float4 ns, uint32_t px,
                                          uint32_t py, bool active, uint64_t active_mask) {
    const f3 pos = {pd.x, pd.y, pd.z};
    // view = normalize(view_position - position) (lib.rs:79-80); normal = normalize(n) (lighting.rs:229)
    float vx = fp.view_position[0] - pos.x, vy = fp.view_position[1] - pos.y, vz = fp.view_position[2] - pos.z;
    float inv_v = rsq(dot3(vx, vy, vz, vx, vy, vz));
    const f3 v = {vx * inv_v, vy * inv_v, vz * inv_v};
    float inv_n = rsq(dot3(ns.x, ns.y, ns.z, ns.x, ns.y, ns.z));
    const f3 n = {ns.x * inv_n, ns.y * inv_n, ns.z * inv_n};
    const float nov_raw = dot3(n.x, n.y, n.z, v.x, v.y, v.z);
    const float nov = fmaxf(nov_raw, kEpsilon);

    light_acc acc = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};

    // sun (lighting.rs:37-53 / 171-177)
    eval_light<TRANSMISSIVE>(acc, m, n, v, nov_raw, nov, {fp.sun_dir[0], fp.sun_dir[1], fp.sun_dir[2]},
                             {fp.sun_intensity[0], fp.sun_intensity[1], fp.sun_intensity[2]});

    // clustered punctual lights (lighting.rs:55-92 / 179-217)
    uint32_t cluster = cluster_index(fp, px, py, pd.w);
    bool in_range = cluster < fp.num_clusters_total;  // out-of-range reads as 0 lights (robust access)
    uint32_t c_safe = in_range ? cluster : 0u;
    int first = __ffsll((unsigned long long)active_mask) - 1;
    uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c_safe, first);
    bool c_uniform = __ballot(active && (c_safe != c0 || !in_range)) == 0ull;
    uint32_t num_lights = 0;
    if (c_uniform) {
        // every live lane reads the same list: scalar loads, lights live in SGPRs
        c0 = __builtin_amdgcn_readfirstlane(c0);
        num_lights = tb.cluster_counts[c0];
        const uint32_t* idx = tb.light_indices + (size_t)c0 * TR_MAX_LIGHTS_PER_CLUSTER;
        for (uint32_t i = 0; i < num_lights; ++i) {
            const tr_dlight& L = tb.lights[idx[i]];
            eval_punctual<TRANSMISSIVE>(acc, m, L, pos, n, v, nov_raw, nov);
        }
    } else {
        num_lights = in_range ? tb.cluster_counts[c_safe] : 0u;
        if (!active) num_lights = 0;
        const uint32_t* idx = tb.light_indices + (size_t)c_safe * TR_MAX_LIGHTS_PER_CLUSTER;
        for (uint32_t i = 0; i < num_lights; ++i) {
            const tr_dlight L = tb.lights[idx[i]];
            eval_punctual<TRANSMISSIVE>(acc, m, L, pos, n, v, nov_raw, nov);
        }
    }

    f3 diffuse = {acc.d.x * m.c_diff[0], acc.d.y * m.c_diff[1], acc.d.z * m.c_diff[2]};

    if constexpr (TRANSMISSIVE) {
        // ibl_volume_refraction (glam-pbr/src/lib.rs:292-354)
        // refract(-v, n, ior) :248-256 ; unit length by construction (Snell), so no re-normalise
        float eta = m.eta;
        float k = fmaf(-eta * eta, fmaf(-nov_raw, nov_raw, 1.0f), 1.0f);
        float cn = fmaf(-eta, nov_raw, fast_sqrt(k));   // eta * n.i + sqrt(k), n.i = -n.v
        float len = m.thickness * ns.w;                 // thickness * model_scale :264
        float ex = fmaf(fmaf(-eta, v.x, -cn * n.x), len, pos.x);
        float ey = fmaf(fmaf(-eta, v.y, -cn * n.y), len, pos.y);
        float ez = fmaf(fmaf(-eta, v.z, -cn * n.z), len, pos.z);
        const float* P = fp.proj_view;                  // column-major
        float cx = fmaf(P[8], ez, fmaf(P[4], ey, fmaf(P[0], ex, P[12])));
        float cy = fmaf(P[9], ez, fmaf(P[5], ey, fmaf(P[1], ex, P[13])));
        float cw = fmaf(P[11], ez, fmaf(P[7], ey, fmaf(P[3], ex, P[15])));
        float tu = fmaf(cx / cw, 0.5f, 0.5f);
        float tv = fmaf(cy / cw, 0.5f, 0.5f);
        float lod = fp.log2_fb_width * m.rough_ior;     // :334-335
        f3 T = sample_pyramid<UNIFORM>(pyramid, fp, tb.levels, tu, tv, lod);
        // apply_volume_attenuation (Beer's law) :275-290
        if (m.flags & 1u) {
            T.x *= fast_exp2(m.neg_atten_log2[0] * len);
            T.y *= fast_exp2(m.neg_atten_log2[1] * len);
            T.z *= fast_exp2(m.neg_atten_log2[2] * len);
        }
        float A, B;
        sample_lut(tb.lut_pairs, fp, m, nov_raw, A, B);
        // (1 - (f0*A + f90*B)) * attenuated * base_colour, summed with the btdf lobes
        float tx = fmaf(1.0f - fmaf(m.f0[0], A, m.f90 * B), T.x, acc.t.x) * m.diffuse[0];
        float ty = fmaf(1.0f - fmaf(m.f0[1], A, m.f90 * B), T.y, acc.t.y) * m.diffuse[1];
        float tz = fmaf(1.0f - fmaf(m.f0[2], A, m.f90 * B), T.z, acc.t.z) * m.diffuse[2];
        // lib.rs:157-159: real = tf * transmission; diffuse = lerp(diffuse, real, tf)
        float tf = m.transmission_factor;
        diffuse.x = fmaf(fmaf(tf, tx, -diffuse.x), tf, diffuse.x);
        diffuse.y = fmaf(fmaf(tf, ty, -diffuse.y), tf, diffuse.y);
        diffuse.z = fmaf(fmaf(tf, tz, -diffuse.z), tf, diffuse.z);
    }

    f3 out = {diffuse.x + acc.s.x + m.emission[0], diffuse.y + acc.s.y + m.emission[1],
              diffuse.z + acc.s.z + m.emission[2]};
    if constexpr (!TRANSMISSIVE) {
        if (fp.debug_clusters != 0u) {  // lib.rs:241-245
            f3 a = debug_colour_for_id(num_lights), b = debug_colour_for_id(cluster);
            out = {fmaf(b.x - 0.5f, 0.025f, a.x), fmaf(b.y - 0.5f, 0.025f, a.y), fmaf(b.z - 0.5f, 0.025f, a.z)};
        }
    }
    return out;
}

// ------------------------------------------------------------------------ the shading kernel
// grid: 1-D over 64x4 tiles of the rect, renumbered per XCD (see top of file); block: 256.
template <bool TRANSMISSIVE, typename OutT /* uint2 = RGBA16F, float4 = RGBA32F */>
__global__ __launch_bounds__(256) void shade_kernel(const tr_frame_params fp, const tr_tables tb,
                                                    const float4* __restrict__ pos_depth,
                                                    const float4* __restrict__ nrm_scale,
                                                    const uint32_t* __restrict__ material_id,
                                                    const uint2* __restrict__ pyramid, OutT* __restrict__ hdr,
                                                    uint2* __restrict__ mip0) {
    // XCD-aware renumbering: hardware block b runs on XCD b % 8; give XCD x the x-th eighth of the tiles.
    const uint32_t ntiles = fp.tiles_x * fp.tiles_y;
    uint32_t b = blockIdx.x;
    const uint32_t per = ntiles >> 3;
    uint32_t tile = (b < per * 8u) ? ((b & 7u) * per + (b >> 3)) : b;
    const uint32_t tyi = tile / fp.tiles_x;
    const uint32_t txi = tile - tyi * fp.tiles_x;

    const uint32_t lane_x = threadIdx.x & 63u, wave_y = threadIdx.x >> 6;
    const uint32_t px = fp.rect_x0 + txi * 64u + lane_x;
    const uint32_t py = fp.rect_y0 + tyi * 4u + wave_y;
    const bool inside = px < fp.rect_x1 && py < fp.rect_y1;
    const size_t pix = (size_t)py * fp.width + px;                                        // colour targets
    const size_t gpix = (size_t)(py - fp.g_origin_y) * fp.g_width + (px - fp.g_origin_x);  // G-buffer planes

    uint32_t mat = TR_NOT_COVERED;
    float4 pd = {0.f, 0.f, 0.f, 0.5f}, ns = {0.f, 0.f, 1.f, 1.f};
    if (inside) {
        mat = material_id[gpix];
        pd = pos_depth[gpix];
        ns = nrm_scale[gpix];
    }
    const bool active = inside && mat != TR_NOT_COVERED;
    const uint64_t amask = __ballot(active);

    if constexpr (!TRANSMISSIVE) {
        // uncovered pixels of the opaque pass get the clear colour (src/main.rs:1592-1601)
        if (inside && !active) {
            if constexpr (sizeof(OutT) == 8) hdr[pix] = pack_rgba16f(0.f, 0.f, 0.f, 1.f);
            else hdr[pix] = OutT{0.f, 0.f, 0.f, 1.f};
            if (mip0) mip0[pix] = pack_rgba16f(0.f, 0.f, 0.f, 1.f);
        }
    }
    if (amask == 0ull) return;

    const int first = __ffsll((unsigned long long)amask) - 1;
    uint32_t m0 = (uint32_t)__builtin_amdgcn_readlane((int)mat, first);
    const bool uniform = __ballot(active && mat != m0) == 0ull;
    f3 out;
    if (uniform) {
        m0 = __builtin_amdgcn_readfirstlane(m0);
        out = shade_pixel<TRANSMISSIVE, true>(fp, tb, tb.dmats[m0], pyramid, pd, ns, px, py, active, amask);
    } else {
        const tr_dmat m = tb.dmats[active ? mat : m0];
        out = shade_pixel<TRANSMISSIVE, false>(fp, tb, m, pyramid, pd, ns, px, py, active, amask);
    }
    if (active) {
        if constexpr (sizeof(OutT) == 8) hdr[pix] = pack_rgba16f(out.x, out.y, out.z, 1.0f);
        else hdr[pix] = OutT{out.x, out.y, out.z, 1.0f};
        if constexpr (!TRANSMISSIVE) {
            if (mip0) mip0[pix] = pack_rgba16f(out.x, out.y, out.z, 1.0f);
        }
    }
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
    const float metallic = mi.metallic_factor, rough = mi.roughness_factor, ior = mi.index_of_refraction;
    const float root = (ior - 1.0f) / (ior + 1.0f);
    const float f0d = root * root;                                   // to_dielectric_f0 :190-195
    const float ior_clamp = fminf(fmaxf(ior * 2.0f - 2.0f, 0.0f), 1.0f);
    const float alpha = rough * rough;
    const float alpha_t = alpha * ior_clamp;                         // ActualRoughness::apply_ior :144-146
    d.a2 = alpha * alpha;
    d.at2 = alpha_t * alpha_t;
    d.f90 = mi.specular_factor + (1.0f - mi.specular_factor) * metallic;  // calculate_combined_f90
    for (int k = 0; k < 3; ++k) {
        float diff = mi.diffuse_factor[k];
        d.diffuse[k] = diff;
        float cd = diff + (0.0f - diff) * metallic;                  // c_diff = lerp(diffuse, 0, metallic)
        d.c_diff[k] = cd * kFrac1Pi;
        float ds = f0d * mi.specular_colour_factor[k] * mi.specular_factor;
        d.f0[k] = ds + (diff - ds) * metallic;                       // calculate_combined_f0
        d.df[k] = d.f90 - d.f0[k];
        d.emission[k] = mi.emissive_factor[k];
    }
    d.eta = 1.0f / ior;
    d.transmission_factor = mi.transmission_factor;
    d.thickness = mi.thickness_factor;
    d.rough_ior = rough * ior_clamp;                                 // PerceptualRoughness::apply_ior :157-159
    const bool has_atten = !(mi.attenuation_distance == __builtin_inff());
    d.flags = has_atten ? 1u : 0u;
    for (int k = 0; k < 3; ++k) {
        float coeff = -logf(mi.attenuation_colour[k]) / mi.attenuation_distance;  // :284
        d.neg_atten_log2[k] = has_atten ? (-coeff) * kLog2e : 0.0f;
    }
    // GGX LUT row (v = perceptual roughness; bilinear, clamp to edge)
    float fh = (float)lut_height;
    float y = rough * fh - 0.5f;
    y = fminf(fmaxf(y, -1.0f), fh);
    float fl = floorf(y);
    d.lut_fy = y - fl;
    int a = (int)fl, mx = (int)lut_height - 1;
    d.lut_row0 = (uint32_t)min(max(a, 0), mx) * lut_stride;
    d.lut_row1 = (uint32_t)min(max(a + 1, 0), mx) * lut_stride;
    d._pad[0] = d._pad[1] = d._pad[2] = 0u;
    out[i] = d;
}

// GGX LUT -> pair table: entry k of a row holds (R,G) of texels clamp(k-2) and clamp(k-1)... see below.
// For the unclamped left tap i0 = floor(u*w - 0.5) in [-1, w], entry k = i0 + 1 holds texel
// clamp(i0) in its low half and clamp(i0 + 1) in its high half: one dword load per row.
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

}  // namespace tr
