// The glam-pbr public API (glam-pbr/src/lib.rs), batched on the device: one thread per element.
//
// basic_brdf, transmission_btdf and ibl_volume_refraction run the passes' own device code — digest_factors,
// eval_light, the pyramid and LUT samplers of tr_kernels.h — with the material digested per element from
// `MaterialParams`, so element i equals what the shading passes compute for a pixel with those inputs.  The small
// functions (d_ggx, v_smith_ggx_correlated, fresnel_schlick, compute_f0, light_direction_and_attenuation) are the
// reference's formulas in its operation order, IEEE division and square root, contraction off.
//
// The arrays are arrays of packed float structs (88 / 76 / 168 bytes per element): neighbouring threads read
// neighbouring elements, every fetched cache line is used in full, and the kernels are HBM bound
// (bytes in + bytes out per element).
#pragma once

#include "tr_kernels.h"

namespace tr {

// MaterialParams -> the per-lane digest the light evaluation reads (the part of tr_dmat that depends on them).
__device__ __forceinline__ void digest_material_params(lane_dmat& lm, const tr_material_params& mp, uint32_t lut_height,
                                                       uint32_t lut_stride) {
#pragma clang fp contract(off)
    const float ior = mp.index_of_refraction;
    const float root = (ior - 1.0f) / (ior + 1.0f);                          // to_dielectric_f0 :190-195
    const float f0_dielectric = root * root;
    const float ior_clamp = fminf(fmaxf(ior * 2.0f - 2.0f, 0.0f), 1.0f);     // :144-146, :157-159
    digest_factors<false>(lm, mp.metallic, mp.perceptual_roughness, ior_clamp, f0_dielectric, mp.specular_factor,
                          mp.specular_colour[0], mp.specular_colour[1], mp.specular_colour[2], mp.diffuse_colour[0],
                          mp.diffuse_colour[1], mp.diffuse_colour[2], lut_height, lut_stride);
    lm.metallic = mp.metallic;
    lm.rough = mp.perceptual_roughness;
    lm.eta = 1.0f / ior;
}

__device__ __forceinline__ void frame_of(pixel_frame& px, const lane_dmat& lm, const float n[3], const float v[3]) {
    px.n = {n[0], n[1], n[2]};
    px.v = {v[0], v[1], v[2]};
    px.nvx = v2f{n[0], v[0]};
    px.nvy = v2f{n[1], v[1]};
    px.nvz = v2f{n[2], v[2]};
    px.nov_raw = dot3(n[0], n[1], n[2], v[0], v[1], v[2]);
    px.nov = fmaxf(px.nov_raw, kEpsilon);
    const v2f ra = pk_fma(splat(px.nov * px.nov), v2f{m_oma2(lm, 0), m_oma2(lm, 1)}, v2f{lm.a2[0], lm.a2[1]});
    px.g_nov = v2f{fast_sqrt(ra.x), fast_sqrt(ra.y)};
}

template <class T>
__device__ __forceinline__ T load_element(const T* array, uint32_t i) {
    static_assert(sizeof(T) % 4 == 0, "packed float records");
    T t;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(array) + (size_t)i * (sizeof(T) / 4);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&t);
#pragma unroll
    for (uint32_t k = 0; k < sizeof(T) / 4; ++k) dst[k] = src[k];
    return t;
}

// basic_brdf (:377-423)
__global__ __launch_bounds__(256) void basic_brdf_kernel(const tr_basic_brdf_params* __restrict__ params, uint32_t count,
                                                         tr_brdf_result* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const tr_basic_brdf_params p = load_element(params, i);
    lane_dmat lm;
    digest_material_params(lm, p.material_params, 1u, 0u);
    pixel_frame px;
    frame_of(px, lm, p.normal, p.view);
    light_acc acc = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const lane_dmat& m = lm;
    eval_light<false>(acc, m, px, {p.light[0], p.light[1], p.light[2]},
                      {p.light_intensity[0], p.light_intensity[1], p.light_intensity[2]}, false);
    float* o = reinterpret_cast<float*>(out) + (size_t)i * 6u;
    o[0] = acc.d.x * mat_c_diff(&lm, 0);
    o[1] = acc.d.y * mat_c_diff(&lm, 1);
    o[2] = acc.d.z * mat_c_diff(&lm, 2);
    o[3] = acc.s.x;
    o[4] = acc.s.y;
    o[5] = acc.s.z;
}

// transmission_btdf (:200-233)
__global__ __launch_bounds__(256) void transmission_btdf_kernel(const tr_transmission_btdf_params* __restrict__ params,
                                                                uint32_t count, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const tr_transmission_btdf_params p = load_element(params, i);
    lane_dmat lm;
    digest_material_params(lm, p.material_params, 1u, 0u);
    pixel_frame px;
    frame_of(px, lm, p.normal, p.view);
    light_acc acc = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const lane_dmat& m = lm;
    eval_light<true>(acc, m, px, {p.light[0], p.light[1], p.light[2]}, {1.0f, 1.0f, 1.0f}, true);
    float* o = out + (size_t)i * 3u;
    o[0] = fmaf(-mat_bt_b(&lm, 0), acc.tb.x, mat_bt_a(&lm, 0) * acc.ta.x) * lm.diffuse[0];
    o[1] = fmaf(-mat_bt_b(&lm, 1), acc.tb.y, mat_bt_a(&lm, 1) * acc.ta.y) * lm.diffuse[1];
    o[2] = fmaf(-mat_bt_b(&lm, 2), acc.tb.z, mat_bt_a(&lm, 2) * acc.ta.z) * lm.diffuse[2];
}

// ibl_volume_refraction (:292-354); the two sampler closures are the opaque pyramid and the GGX LUT.
struct tr_ibl_tables {
    const uint2* pyramid;
    const tr_level_table* levels;
    uint32_t pyr_levels;
    const uint32_t* lut_pairs;
    uint32_t lut_width, lut_height, lut_stride;
};

// The two halves the function's closures cut it into.  ibl_request: everything up to the closures' arguments (:326-341);
// ibl_finish: everything behind their answers (:338-353).
struct ibl_requests {
    float tu, tv, lod;     // framebuffer_sampler(texture_coords, framebuffer_lod)
    float nov, rough;      // ggx_lut_sampler(normal_dot_view, perceptual_roughness)
    float len;             // ray_length (apply_volume_attenuation's transmission_distance)
};
__device__ __forceinline__ ibl_requests ibl_request(const tr_ibl_volume_refraction_params& p, const lane_dmat& lm) {
    const float* n = p.normal;
    const float* v = p.view;
    ibl_requests q;
    q.nov = dot3(n[0], n[1], n[2], v[0], v[1], v[2]);
    // refract(-v, n, ior) :248-256, unit length by construction; ray = that * thickness * model_scale :258-268
    const float eta = lm.eta;
    const float k = fmaf(-eta * eta, fmaf(-q.nov, q.nov, 1.0f), 1.0f);
    const float cn = fmaf(-eta, q.nov, fast_sqrt(k));
    q.len = p.thickness * p.model_scale;
    const float ex = fmaf(fmaf(-eta, v[0], -cn * n[0]), q.len, p.position[0]);
    const float ey = fmaf(fmaf(-eta, v[1], -cn * n[1]), q.len, p.position[1]);
    const float ez = fmaf(fmaf(-eta, v[2], -cn * n[2]), q.len, p.position[2]);
    const float* P = p.proj_view_matrix;
    const float cx = fmaf(P[8], ez, fmaf(P[4], ey, fmaf(P[0], ex, P[12])));
    const float cy = fmaf(P[9], ez, fmaf(P[5], ey, fmaf(P[1], ex, P[13])));
    const float cw = fmaf(P[11], ez, fmaf(P[7], ey, fmaf(P[3], ex, P[15])));
    const float hw = 0.5f * rcp(cw);
    q.tu = fmaf(cx, hw, 0.5f);
    q.tv = fmaf(cy, hw, 0.5f);
    q.lod = fast_log2((float)p.framebuffer_size_x) * m_rough_ior(lm);   // :334-335
    q.rough = lm.rough;
    return q;
}
__device__ __forceinline__ void ibl_finish(const tr_ibl_volume_refraction_params& p, const lane_dmat& lm, f3 T, v2f AB, float len,
                                           float* o) {
    if (!(p.attenuation_distance == __builtin_inff())) {   // apply_volume_attenuation :275-290
        const float att[3] = {p.attenuation_colour[0], p.attenuation_colour[1], p.attenuation_colour[2]};
        float tr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float neg_coeff_log2 = (logf(att[c]) / p.attenuation_distance) * kLog2e;   // -(-ln(c) / d) * log2(e)
            tr[c] = fast_exp2(neg_coeff_log2 * len);
        }
        T.x *= tr[0];
        T.y *= tr[1];
        T.z *= tr[2];
    }
    const float fb = lm.f90 * AB.y;
    o[0] = (1.0f - fmaf(lm.f0[0], AB.x, fb)) * T.x * lm.diffuse[0];
    o[1] = (1.0f - fmaf(lm.f0[1], AB.x, fb)) * T.y * lm.diffuse[1];
    o[2] = (1.0f - fmaf(lm.f0[2], AB.x, fb)) * T.z * lm.diffuse[2];
}

__global__ __launch_bounds__(256) void ibl_volume_refraction_kernel(const tr_ibl_volume_refraction_params* __restrict__ params,
                                                                    uint32_t count, const tr_ibl_tables t,
                                                                    float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = i < count;
    // (every lane of the wave runs the sampler: it splits the wave by mip level with ballots; the tail reads element 0)
    const tr_ibl_volume_refraction_params p = load_element(params, live ? i : 0u);
    lane_dmat lm;
    digest_material_params(lm, p.material_params, t.lut_height, t.lut_stride);
    const ibl_requests q = ibl_request(p, lm);
    pyramid_fetch pf;
    pyramid_issue(pf, t.pyramid, as_constant(t.levels), t.pyr_levels, q.tu, q.tv, q.lod, lane);
    lut_fetch lf;
    uint32_t row0, row1;
    lut_rows(lm.rough, t.lut_height, t.lut_stride, lf.fy, row0, row1);
    lut_issue(lf, t.lut_pairs, (float)t.lut_width, row0, row1, q.nov);
    const f3 T = pyramid_resolve(pf);
    const v2f AB = lut_resolve(lf, lf.fy);
    if (!live) return;
    ibl_finish(p, lm, T, AB, q.len, out + (size_t)i * 3u);
}

// ibl_volume_refraction<FSamp, GSamp> with the CALLER's closures: the requests, then the rest given their answers.
__global__ __launch_bounds__(256) void ibl_requests_kernel(const tr_ibl_volume_refraction_params* __restrict__ params, uint32_t count,
                                                           float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const tr_ibl_volume_refraction_params p = load_element(params, i);
    lane_dmat lm;
    digest_material_params(lm, p.material_params, 1u, 0u);
    const ibl_requests q = ibl_request(p, lm);
    float* o = out + (size_t)i * 5u;
    o[0] = q.tu; o[1] = q.tv; o[2] = q.lod; o[3] = q.nov; o[4] = q.rough;
}
__global__ __launch_bounds__(256) void ibl_resolve_kernel(const tr_ibl_volume_refraction_params* __restrict__ params, uint32_t count,
                                                          const float* __restrict__ framebuffer_rgb, const float* __restrict__ lut_ab,
                                                          float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const tr_ibl_volume_refraction_params p = load_element(params, i);
    lane_dmat lm;
    digest_material_params(lm, p.material_params, 1u, 0u);
    const f3 T = {framebuffer_rgb[(size_t)i * 3u], framebuffer_rgb[(size_t)i * 3u + 1u], framebuffer_rgb[(size_t)i * 3u + 2u]};
    const v2f AB = {lut_ab[(size_t)i * 2u], lut_ab[(size_t)i * 2u + 1u]};
    ibl_finish(p, lm, T, AB, p.thickness * p.model_scale, out + (size_t)i * 3u);
}

// light_direction_and_attenuation (:12-23)
__global__ __launch_bounds__(256) void light_direction_kernel(const float* __restrict__ fragment_position,
                                                              const float* __restrict__ light_position, uint32_t count,
                                                              float* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float* f = fragment_position + (size_t)i * 3u;
    const float* l = light_position + (size_t)i * 3u;
    const float dx = l[0] - f[0], dy = l[1] - f[1], dz = l[2] - f[2];
    const float distance_sq = (dx * dx + dy * dy) + dz * dz;
    const float distance = __fsqrt_rn(distance_sq);
    float* o = out + (size_t)i * 5u;
    o[0] = dx / distance;
    o[1] = dy / distance;
    o[2] = dz / distance;
    o[3] = distance;
    o[4] = 1.0f / distance_sq;
}

// d_ggx (:101-109)
__global__ __launch_bounds__(256) void d_ggx_kernel(const float* __restrict__ noh, const float* __restrict__ roughness,
                                                    uint32_t count, float* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float a2 = roughness[i] * roughness[i];
    const float f = (noh[i] * noh[i]) * (a2 - 1.0f) + 1.0f;
    out[i] = a2 / (kPi * f * f);
}

// v_smith_ggx_correlated (:114-133)
__global__ __launch_bounds__(256) void v_smith_kernel(const float* __restrict__ nov_, const float* __restrict__ nol_,
                                                      const float* __restrict__ roughness, uint32_t count,
                                                      float* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float nov = nov_[i], nol = nol_[i], a2 = roughness[i] * roughness[i];
    const float ggx_v = nol * __fsqrt_rn(nov * nov * (1.0f - a2) + a2);
    const float ggx_l = nov * __fsqrt_rn(nol * nol * (1.0f - a2) + a2);
    const float ggx = ggx_v + ggx_l;
    out[i] = ggx > 0.0f ? 0.5f / ggx : 0.0f;
}

// fresnel_schlick (:137-139); (1 - v.h)^5 by multiplication (the reference calls powf(x, 5.0))
__global__ __launch_bounds__(256) void fresnel_schlick_kernel(const float* __restrict__ voh, const float* __restrict__ f0,
                                                              const float* __restrict__ f90, uint32_t count,
                                                              float* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float x = 1.0f - voh[i], x2 = x * x, p = x2 * x2 * x;
#pragma unroll
    for (uint32_t c = 0; c < 3u; ++c) {
        const float a = f0[(size_t)i * 3u + c], b = f90[(size_t)i * 3u + c];
        out[(size_t)i * 3u + c] = a + (b - a) * p;
    }
}

// compute_f0 (:454-465)
__global__ __launch_bounds__(256) void compute_f0_kernel(const float* __restrict__ metallic, const float* __restrict__ ior_,
                                                         const float* __restrict__ diffuse, uint32_t count,
                                                         float* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float ior = ior_[i], m = metallic[i];
    const float root = (ior - 1.0f) / (ior + 1.0f);
    const float dielectric = (1.0f - m) * (root * root);
#pragma unroll
    for (uint32_t c = 0; c < 3u; ++c) out[(size_t)i * 3u + c] = dielectric + m * diffuse[(size_t)i * 3u + c];
}

}  // namespace tr
