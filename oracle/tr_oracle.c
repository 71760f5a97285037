/*
 * tr_oracle.c — see tr_oracle.h.  TEST INFRASTRUCTURE: the checker, never the product.
 *
 * Scalar IEEE fp32 (`real` = float; see tr_oracle.h for the fp64 conditioning twin), one operation per
 * source operation of the reference, in the
 * reference's association order (glam 0.19 scalar Vec3: dot = (x*x'+y*y')+z*z',
 * normalize = v * (1/sqrt(len^2)), lerp = a + (b-a)*t; confirmed against the op order of
 * compiled-shaders/normal/fragment_transmission.spv by oracle/spirv_ref).
 * Must be compiled with -ffp-contract=off and without -ffast-math.
 */
#include "tr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(O_REAL_DOUBLE)
#define R(x) x
#define R_SQRT sqrt
#define R_POW pow
#define R_LOG log
#define R_EXP exp
#define R_LOG2 log2
#define R_FLOOR floor
#define R_MAX fmax
#define R_MIN fmin
#define R_COS cos
#define R_SIN sin
#define R_TAN tan
#define R_ABS fabs
#else
#define R(x) x##f
#define R_SQRT sqrtf
#define R_POW powf
#define R_LOG logf
#define R_EXP expf
#define R_LOG2 log2f
#define R_FLOOR floorf
#define R_MAX fmaxf
#define R_MIN fminf
#define R_COS cosf
#define R_ABS fabsf
#define R_SIN sinf
#define R_TAN tanf
#endif

/* ------------------------------------------------------------------ vec3 */
static inline o_vec3 v3(real x, real y, real z) { o_vec3 r = {x, y, z}; return r; }
static inline o_vec3 v3_splat(real s) { return v3(s, s, s); }
static inline o_vec3 v3_add(o_vec3 a, o_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline o_vec3 v3_sub(o_vec3 a, o_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline o_vec3 v3_mul(o_vec3 a, o_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline o_vec3 v3_scale(o_vec3 a, real s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline o_vec3 v3_div(o_vec3 a, real s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline o_vec3 v3_neg(o_vec3 a) { return v3(-a.x, -a.y, -a.z); }
static inline o_vec3 v3_add_s(o_vec3 a, real s) { return v3(a.x + s, a.y + s, a.z + s); }
/* 1.0 - v */
static inline o_vec3 v3_one_minus(o_vec3 a) { return v3(R(1.0) - a.x, R(1.0) - a.y, R(1.0) - a.z); }
static inline real v3_dot(o_vec3 a, o_vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline o_vec3 v3_normalize(o_vec3 a) {
    real len = R_SQRT(v3_dot(a, a));
    real recip = R(1.0) / len;
    return v3_scale(a, recip);
}
static inline o_vec3 v3_lerp(o_vec3 a, o_vec3 b, real t) { return v3_add(a, v3_scale(v3_sub(b, a), t)); }
static inline real v3_max_element(o_vec3 a) { return R_MAX(a.x, R_MAX(a.y, a.z)); }

/* glam-pbr/src/lib.rs:25-27 */
static inline real o_clamp(real value, real mn, real mx) { return R_MIN(R_MAX(value, mn), mx); }

/* Rust `f32 as u32`: saturating, NaN -> 0 */
static inline uint32_t f32_as_u32(real f) {
    if (!(f > R(0.0))) return 0u;
    if (f >= R(4294967296.0)) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

#define O_EPSILON R(1.1920929e-07) /* core::f32::EPSILON */
#define O_PI R(3.14159265358979323846)
#define O_FRAC_1_PI R(0.318309886183790671538)

/* ------------------------------------------------------------- glam-pbr */

/* glam-pbr/src/lib.rs:12-23 */
void o_light_direction_and_attenuation(o_vec3 fragment_position, o_vec3 light_position,
                                       o_vec3* direction, real* distance, real* attenuation) {
    o_vec3 vector = v3_sub(light_position, fragment_position);
    real distance_sq = v3_dot(vector, vector);
    real dist = R_SQRT(distance_sq);
    *direction = v3_div(vector, dist);
    *distance = dist;
    *attenuation = R(1.0) / distance_sq;
}

/* glam-pbr/src/lib.rs:93-98 `Dot::new` */
real o_dot_clamped(o_vec3 a, o_vec3 b) { return R_MAX(v3_dot(a, b), O_EPSILON); }

/* glam-pbr/src/lib.rs:62-68 `Halfway::new` */
static inline o_vec3 o_halfway(o_vec3 view, o_vec3 light) { return v3_normalize(v3_add(view, light)); }

/* glam-pbr/src/lib.rs:101-109 */
real o_d_ggx(real noh, real actual_roughness) {
    real alpha_roughness_sq = actual_roughness * actual_roughness;
    real f = (noh * noh) * (alpha_roughness_sq - R(1.0)) + R(1.0);
    return alpha_roughness_sq / (O_PI * f * f);
}

/* glam-pbr/src/lib.rs:114-133 */
real o_v_smith_ggx_correlated(real nov, real nol, real actual_roughness) {
    real a2 = actual_roughness * actual_roughness;
    real ggx_v = nol * R_SQRT(nov * nov * (R(1.0) - a2) + a2);
    real ggx_l = nov * R_SQRT(nol * nol * (R(1.0) - a2) + a2);
    real ggx = ggx_v + ggx_l;
    if (ggx > R(0.0)) return R(0.5) / ggx;
    return R(0.0);
}

/* glam-pbr/src/lib.rs:137-139 */
o_vec3 o_fresnel_schlick(real voh, o_vec3 f0, o_vec3 f90) {
    real p = R_POW(R(1.0) - voh, R(5.0));
    return v3_add(f0, v3_scale(v3_sub(f90, f0), p));
}

/* glam-pbr/src/lib.rs:190-195 */
real o_to_dielectric_f0(real ior) {
    real root = (ior - R(1.0)) / (ior + R(1.0));
    return root * root;
}

/* glam-pbr/src/lib.rs:425-430 */
static o_vec3 o_calculate_combined_f0(o_material_params m) {
    o_vec3 dielectric_specular_f0 =
        v3_scale(v3_scale(m.specular_colour, o_to_dielectric_f0(m.index_of_refraction)), m.specular_factor);
    return v3_lerp(dielectric_specular_f0, m.diffuse_colour, m.metallic);
}

/* glam-pbr/src/lib.rs:432-435 */
static o_vec3 o_calculate_combined_f90(o_material_params m) {
    return v3_lerp(v3_splat(m.specular_factor), v3_splat(R(1.0)), m.metallic);
}

/* glam-pbr/src/lib.rs:200-233 */
o_vec3 o_transmission_btdf(o_material_params m, o_vec3 normal, o_vec3 view, o_vec3 light) {
    real actual_roughness = m.perceptual_roughness * m.perceptual_roughness;             /* :149-151 */
    real ior = m.index_of_refraction;
    real transmission_roughness = actual_roughness * o_clamp(ior * R(2.0) - R(2.0), R(0.0), R(1.0)); /* :144-146 */

    /* light.0 + 2.0 * normal.0 * (-light.0).dot(normal.0) */
    real d = v3_dot(v3_neg(light), normal);
    o_vec3 light_mirrored = v3_normalize(v3_add(light, v3_scale(v3_scale(normal, R(2.0)), d)));

    o_vec3 halfway = o_halfway(view, light_mirrored);
    real noh = o_dot_clamped(normal, halfway);
    real voh = o_dot_clamped(view, halfway);
    real nov = o_dot_clamped(normal, view);
    real nol_m = o_dot_clamped(normal, light_mirrored);

    real distribution = o_d_ggx(noh, transmission_roughness);
    real geometric_shadowing = o_v_smith_ggx_correlated(nov, nol_m, transmission_roughness);

    o_vec3 f0 = o_calculate_combined_f0(m);
    o_vec3 f90 = o_calculate_combined_f90(m);
    o_vec3 fresnel = o_fresnel_schlick(voh, f0, f90);

    /* (1.0 - fresnel) * distribution * geometric_shadowing * diffuse_colour */
    return v3_mul(v3_scale(v3_scale(v3_one_minus(fresnel), distribution), geometric_shadowing), m.diffuse_colour);
}

/* glam-pbr/src/lib.rs:248-256 */
o_vec3 o_refract(o_vec3 incident, o_vec3 normal, real ior) {
    real eta = R(1.0) / ior;
    real n_dot_i = v3_dot(normal, incident);
    real k = R(1.0) - eta * eta * (R(1.0) - n_dot_i * n_dot_i);
    return v3_sub(v3_scale(incident, eta), v3_scale(normal, eta * n_dot_i + R_SQRT(k)));
}

/* glam-pbr/src/lib.rs:275-290 */
o_vec3 o_apply_volume_attenuation(o_vec3 transmitted_light, real transmission_distance,
                                  real attenuation_distance, o_vec3 attenuation_colour) {
    if (attenuation_distance == INFINITY) return transmitted_light;
    o_vec3 lnc = v3(R_LOG(attenuation_colour.x), R_LOG(attenuation_colour.y), R_LOG(attenuation_colour.z));
    o_vec3 attenuation_coefficient = v3_div(v3_neg(lnc), attenuation_distance);
    o_vec3 e = v3_scale(v3_neg(attenuation_coefficient), transmission_distance);
    o_vec3 transmittance = v3(R_EXP(e.x), R_EXP(e.y), R_EXP(e.z));
    return v3_mul(transmittance, transmitted_light);
}

/* glam 0.19 Mat4 * Vec4: ((X*x + Y*y) + Z*z) + W*w per component, column-major m */
static inline void mat4_mul_vec4(const real m[16], const real v[4], real out[4]) {
    for (int r = 0; r < 4; ++r) {
        real acc = m[0 + r] * v[0];
        acc = m[4 + r] * v[1] + acc;
        acc = m[8 + r] * v[2] + acc;
        acc = m[12 + r] * v[3] + acc;
        out[r] = acc;
    }
}

/* glam-pbr/src/lib.rs:292-354 */
o_vec3 o_ibl_volume_refraction(const o_ibl_volume_refraction_params* p,
                               o_framebuffer_sampler fb, void* fb_user,
                               o_ggx_lut_sampler lut, void* lut_user) {
    o_material_params m = p->material_params;
    real ior = m.index_of_refraction;

    /* get_volume_transmission_ray :258-268 */
    o_vec3 refraction = o_refract(v3_neg(p->view), p->normal, ior);
    real ray_length = p->thickness * p->model_scale;
    o_vec3 ray = v3_scale(v3_normalize(refraction), ray_length);
    o_vec3 refracted_ray_exit = v3_add(p->position, ray);

    real e4[4] = {refracted_ray_exit.x, refracted_ray_exit.y, refracted_ray_exit.z, R(1.0)};
    real dc[4];
    mat4_mul_vec4(p->proj_view_matrix, e4, dc);
    o_vec2 screen = {dc[0] / dc[3], dc[1] / dc[3]};
    o_vec2 tex = {(screen.x + R(1.0)) / R(2.0), (screen.y + R(1.0)) / R(2.0)};

    /* (framebuffer_size_x as f32).log2() * perceptual_roughness.apply_ior(ior).0 :334-335 */
    real rough_ior = m.perceptual_roughness * o_clamp(ior * R(2.0) - R(2.0), R(0.0), R(1.0)); /* :157-159 */
    real framebuffer_lod = R_LOG2((real)p->framebuffer_size_x) * rough_ior;

    o_vec3 transmitted_light = fb(fb_user, tex, framebuffer_lod);
    o_vec3 attenuated = o_apply_volume_attenuation(transmitted_light, ray_length,
                                                   p->attenuation_distance, p->attenuation_colour);

    real normal_dot_view = v3_dot(p->normal, p->view); /* unclamped :345 */
    o_vec2 brdf = lut(lut_user, normal_dot_view, m.perceptual_roughness);

    o_vec3 f0 = o_calculate_combined_f0(m);
    o_vec3 f90 = o_calculate_combined_f90(m);
    o_vec3 specular_colour = v3_add(v3_scale(f0, brdf.x), v3_scale(f90, brdf.y));

    return v3_mul(v3_mul(v3_one_minus(specular_colour), attenuated), m.diffuse_colour);
}

/* glam-pbr/src/lib.rs:377-423 (diffuse_brdf :356-360, specular_brdf :362-375 inlined) */
o_brdf_result o_basic_brdf(o_vec3 normal, o_vec3 light, o_vec3 light_intensity, o_vec3 view,
                           o_material_params m) {
    real actual_roughness = m.perceptual_roughness * m.perceptual_roughness;

    o_vec3 halfway = o_halfway(view, light);
    real noh = o_dot_clamped(normal, halfway);
    real nov = o_dot_clamped(normal, view);
    real nol = o_dot_clamped(normal, light);
    real voh = o_dot_clamped(view, halfway);

    o_vec3 c_diff = v3_lerp(m.diffuse_colour, v3_splat(R(0.0)), m.metallic);

    o_vec3 f0 = o_calculate_combined_f0(m);
    o_vec3 f90 = o_calculate_combined_f90(m);
    o_vec3 fresnel = o_fresnel_schlick(voh, f0, f90);

    /* diffuse_brdf: (1.0 - fresnel.max_element()) * FRAC_1_PI * base */
    o_vec3 dbrdf = v3_scale(c_diff, (R(1.0) - v3_max_element(fresnel)) * O_FRAC_1_PI);
    /* specular_brdf: (D * V) * fresnel */
    real dv = o_d_ggx(noh, actual_roughness) * o_v_smith_ggx_correlated(nov, nol, actual_roughness);
    o_vec3 sbrdf = v3_scale(fresnel, dv);

    o_brdf_result r;
    r.diffuse = v3_mul(v3_scale(light_intensity, nol), dbrdf);
    r.specular = v3_mul(v3_scale(light_intensity, nol), sbrdf);
    return r;
}

/* glam-pbr/src/lib.rs:454-465 */
o_vec3 o_compute_f0(real metallic, real ior, o_vec3 diffuse_colour) {
    real dielectric_f0 = o_to_dielectric_f0(ior);
    return v3_add(v3_splat((R(1.0) - metallic) * dielectric_f0), v3_scale(diffuse_colour, metallic));
}

/* ---------------------------------------------------------- shared-structs */

/* shared-structs/src/lib.rs:44-52 */
void o_light_cluster_coefficients_new(real z_near, real z_far, uint32_t slices,
                                      tr_light_cluster_coefficients* out) {
    memset(out, 0, sizeof(*out));
    out->z_near = z_near;
    out->z_far = z_far;
    out->num_depth_slices = slices;
    out->scale = (real)slices / R_LOG2(z_far / z_near);
    out->bias = -((real)slices * R_LOG2(z_near) / R_LOG2(z_far / z_near));
}

/* shared-structs/src/lib.rs:54-63.  Index work: evaluated in fp32 in BOTH builds of this file — the fp64 twin
 * exists to show how much of a pixel's VALUE is rounding noise of the reference's formulas, and must read the
 * same light list as the reference to do so. */
uint32_t o_get_depth_slice(const tr_light_cluster_coefficients* c, real frag_depth_in) {
    float frag_depth = (float)frag_depth_in;
    float depth_range = 2.0f * (1.0f - frag_depth) - 1.0f;
    float linear = 2.0f * c->z_near * c->z_far / (c->z_far + c->z_near - depth_range * (c->z_far - c->z_near));
    return f32_as_u32((real)fmaxf(log2f(linear) * c->scale + c->bias, 0.0f));
}

/* shared-structs/src/lib.rs:129-138 */
real o_spotlight_factor(const tr_light* l, o_vec3 direction_to_light) {
    o_vec3 spot_dir = v3(l->spotlight_direction_and_outer_angle[0], l->spotlight_direction_and_outer_angle[1],
                         l->spotlight_direction_and_outer_angle[2]);
    real theta = v3_dot(v3_neg(direction_to_light), spot_dir);
    real outer_angle = l->spotlight_direction_and_outer_angle[3];
    real epsilon = l->position_and_spotlight_epsilon[3];
    return R_MAX((theta - R_COS(outer_angle)) / epsilon, R(0.0));
}

/* ------------------------------------------------------------ host helpers */

/* src/main.rs:2590-2592 */
uint32_t o_mip_levels_for_size(uint32_t w, uint32_t h) {
    uint32_t m = w < h ? w : h;
    return f32_as_u32(R_LOG2((real)m)) + 1u;
}

/* src/main.rs:39-54 */
void o_perspective_matrix_reversed(uint32_t w, uint32_t h, real out[16]) {
    const real z_near = R(0.01), z_far = R(500.0);
    real aspect_ratio = (real)w / (real)h;
    real vertical_fov = R(59.0) * (O_PI / R(180.0));
    real focal_length = R(1.0) / R_TAN(vertical_fov / R(2.0));
    real a = z_near / (z_far - z_near);
    real b = z_far * a;
    memset(out, 0, 16 * sizeof(real));
    out[0] = focal_length / aspect_ratio;
    out[5] = -focal_length;
    out[10] = a;
    out[11] = -R(1.0);
    out[14] = b;
}

/* src/main.rs:2715-2722 */
void o_sun_as_normal(real pitch, real yaw, real out[3]) {
    out[0] = R_COS(pitch) * R_SIN(yaw);
    out[1] = R_SIN(pitch);
    out[2] = R_COS(pitch) * R_COS(yaw);
}

/* ------------------------------------------------------------------- half */

uint16_t o_f32_to_f16(real fr) {
    float f = (float)fr;
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007FFFFFu;
    int32_t exp = (int32_t)((x >> 23) & 0xFF);
    if (exp == 0xFF) return (uint16_t)(sign | 0x7C00u | (mant ? (0x0200u | (mant >> 13)) : 0u));
    int32_t e = exp - 127 + 15;
    if (e >= 0x1F) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t half_mant = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1u);
        if (rem > halfway || (rem == halfway && (half_mant & 1u))) half_mant++;
        return (uint16_t)(sign | half_mant);
    }
    uint32_t half = ((uint32_t)e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++; /* may carry into exponent -> inf: correct */
    return (uint16_t)(sign | half);
}

real o_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t mant = h & 0x3FFu;
    uint32_t x;
    if (exp == 0) {
        if (mant == 0) {
            x = sign;
        } else {
            int e = -1;
            do { e++; mant <<= 1; } while (!(mant & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3FFu) << 13);
        }
    } else if (exp == 0x1F) {
        x = sign | 0x7F800000u | (mant << 13);
    } else {
        x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return (real)f;
}

/* --------------------------------------------------------------- sampling */

void o_pyramid_layout(uint32_t w, uint32_t h, o_pyramid* out, uint64_t* total_texels) {
    memset(out, 0, sizeof(*out));
    out->width = w;
    out->height = h;
    out->levels = o_mip_levels_for_size(w, h);
    if (out->levels > TR_MAX_MIP_LEVELS) out->levels = TR_MAX_MIP_LEVELS;
    uint64_t off = 0;
    for (uint32_t l = 0; l < out->levels; ++l) {
        uint32_t lw = (w >> l) ? (w >> l) : 1u, lh = (h >> l) ? (h >> l) : 1u;
        out->level_offset[l] = (uint32_t)off;
        off += (uint64_t)lw * lh;
    }
    if (total_texels) *total_texels = off;
}

static inline uint32_t level_dim(uint32_t d, uint32_t l) { return (d >> l) ? (d >> l) : 1u; }

/* One bilinear tap of one level.  Vulkan "texel filtering", VK_FILTER_LINEAR, CLAMP_TO_EDGE:
 * x = u*w - 0.5, i0 = floor(x), i1 = i0+1 (both clamped), weight frac(x).  (unpinned: the
 * spec allows >= 8-bit fixed-point weights; restated with exact fp32 weights, lerp form.) */
static void bilinear_index(real coord, uint32_t dim, uint32_t* i0, uint32_t* i1, real* frac) {
    real x = coord * (real)dim - R(0.5);
    /* clamp to [-1, dim]: identical result for every finite coord, defined for NaN/inf */
    x = R_MIN(R_MAX(x, -R(1.0)), (real)dim);
    real fl = R_FLOOR(x);
    *frac = x - fl;
    int32_t a = (int32_t)fl;
    int32_t b = a + 1;
    int32_t mx = (int32_t)dim - 1;
    if (a < 0) a = 0;
    if (a > mx) a = mx;
    if (b < 0) b = 0;
    if (b > mx) b = mx;
    *i0 = (uint32_t)a;
    *i1 = (uint32_t)b;
}

static o_vec3 pyramid_bilinear(const o_pyramid* p, uint32_t level, real u, real v) {
    uint32_t w = level_dim(p->width, level), h = level_dim(p->height, level);
    const uint16_t* base = p->texels + (size_t)p->level_offset[level] * 4u;
    uint32_t x0, x1, y0, y1;
    real fx, fy;
    bilinear_index(u, w, &x0, &x1, &fx);
    bilinear_index(v, h, &y0, &y1, &fy);
    real c[3];
    for (int k = 0; k < 3; ++k) {
        real t00 = o_f16_to_f32(base[((size_t)y0 * w + x0) * 4u + k]);
        real t10 = o_f16_to_f32(base[((size_t)y0 * w + x1) * 4u + k]);
        real t01 = o_f16_to_f32(base[((size_t)y1 * w + x0) * 4u + k]);
        real t11 = o_f16_to_f32(base[((size_t)y1 * w + x1) * 4u + k]);
        real top = t00 + (t10 - t00) * fx;
        real bot = t01 + (t11 - t01) * fx;
        c[k] = top + (bot - top) * fy;
    }
    return v3(c[0], c[1], c[2]);
}

/* framebuffer.sample_by_lod(clamp_sampler, uv, lod) (shader/src/lib.rs:135-138); sampler
 * state src/main.rs:694-705: LINEAR/LINEAR, mip LINEAR, CLAMP_TO_EDGE, lod in [0, levels-1]. */
o_vec3 o_sample_pyramid(const o_pyramid* p, real u, real v, real lod) {
    real max_lod = (real)(p->levels - 1u);
    real l = R_MIN(R_MAX(lod, R(0.0)), max_lod);
    real lf = R_FLOOR(l);
    real t = l - lf;
    uint32_t l0 = (uint32_t)lf;
    uint32_t l1 = l0 + 1u < p->levels ? l0 + 1u : p->levels - 1u;
    o_vec3 a = pyramid_bilinear(p, l0, u, v);
    o_vec3 b = pyramid_bilinear(p, l1, u, v);
    return v3_add(a, v3_scale(v3_sub(b, a), t));
}

/* textures[ggx_lut].sample(clamp_sampler, (n.v, roughness)).xy (shader/src/lib.rs:126-133):
 * single level R8G8B8A8_UNORM (src/main.rs:316), bilinear, clamp-to-edge, texel = byte/255. */
o_vec2 o_sample_lut(const uint8_t* rgba8, uint32_t w, uint32_t h, real u, real v) {
    uint32_t x0, x1, y0, y1;
    real fx, fy;
    bilinear_index(u, w, &x0, &x1, &fx);
    bilinear_index(v, h, &y0, &y1, &fy);
    real c[2];
    for (int k = 0; k < 2; ++k) {
        real t00 = (real)rgba8[((size_t)y0 * w + x0) * 4u + k] / R(255.0);
        real t10 = (real)rgba8[((size_t)y0 * w + x1) * 4u + k] / R(255.0);
        real t01 = (real)rgba8[((size_t)y1 * w + x0) * 4u + k] / R(255.0);
        real t11 = (real)rgba8[((size_t)y1 * w + x1) * 4u + k] / R(255.0);
        real top = t00 + (t10 - t00) * fx;
        real bot = t01 + (t11 - t01) * fx;
        c[k] = top + (bot - top) * fy;
    }
    o_vec2 r = {c[0], c[1]};
    return r;
}

/* generate_mips (call site src/main.rs:2054-2063; body in the un-vendored
 * ash-opinionated-abstractions@8591c309 => unpinned).  Restated as the vkCmdBlitImage
 * VK_FILTER_LINEAR chain it presumably is: level l from level l-1, whole-image regions,
 * dst texel (i,j) samples the source at unnormalised u = (i+0.5)*(ws/wd), linear filter,
 * clamp to edge; 2x2 box when the source size is even.  fp32 weights/accumulation in the
 * order (t00*w00 + t10*w10) + (t01*w01 + t11*w11), RTNE store, all four channels. */
void o_generate_mips(const o_pyramid* p, uint16_t* texels) {
    for (uint32_t l = 1; l < p->levels; ++l) {
        uint32_t ws = level_dim(p->width, l - 1), hs = level_dim(p->height, l - 1);
        uint32_t wd = level_dim(p->width, l), hd = level_dim(p->height, l);
        const uint16_t* src = texels + (size_t)p->level_offset[l - 1] * 4u;
        uint16_t* dst = texels + (size_t)p->level_offset[l] * 4u;
        real sx = (real)ws / (real)wd, sy = (real)hs / (real)hd;
        for (uint32_t j = 0; j < hd; ++j) {
            real y = ((real)j + R(0.5)) * sy - R(0.5);
            real fy0 = R_FLOOR(y);
            real by = y - fy0;
            int32_t y0 = (int32_t)fy0, y1 = y0 + 1;
            if (y0 < 0) y0 = 0;
            if (y0 > (int32_t)hs - 1) y0 = (int32_t)hs - 1;
            if (y1 > (int32_t)hs - 1) y1 = (int32_t)hs - 1;
            for (uint32_t i = 0; i < wd; ++i) {
                real x = ((real)i + R(0.5)) * sx - R(0.5);
                real fx0 = R_FLOOR(x);
                real ax = x - fx0;
                int32_t x0 = (int32_t)fx0, x1 = x0 + 1;
                if (x0 < 0) x0 = 0;
                if (x0 > (int32_t)ws - 1) x0 = (int32_t)ws - 1;
                if (x1 > (int32_t)ws - 1) x1 = (int32_t)ws - 1;
                real w00 = (R(1.0) - ax) * (R(1.0) - by), w10 = ax * (R(1.0) - by);
                real w01 = (R(1.0) - ax) * by, w11 = ax * by;
                for (int k = 0; k < 4; ++k) {
                    real t00 = o_f16_to_f32(src[((size_t)y0 * ws + (uint32_t)x0) * 4u + k]);
                    real t10 = o_f16_to_f32(src[((size_t)y0 * ws + (uint32_t)x1) * 4u + k]);
                    real t01 = o_f16_to_f32(src[((size_t)y1 * ws + (uint32_t)x0) * 4u + k]);
                    real t11 = o_f16_to_f32(src[((size_t)y1 * ws + (uint32_t)x1) * 4u + k]);
                    real r = (t00 * w00 + t10 * w10) + (t01 * w01 + t11 * w11);
                    dst[((size_t)j * wd + i) * 4u + k] = o_f32_to_f16(r);
                }
            }
        }
    }
}

/* --------------------------------------------------- fragment entry points */

static o_vec3 f3(const float* p) { return v3(p[0], p[1], p[2]); }

/* ------------------------------------------------- clustered-light build */

/* shared-structs/src/lib.rs:65-67 `slice_to_depth` */
real o_slice_to_depth(const tr_light_cluster_coefficients* c, uint32_t slice) {
    return -(real)c->z_near * R_POW((real)c->z_far / (real)c->z_near, (real)slice / (real)c->num_depth_slices);
}

/* shader/src/lib.rs:582-594 `line_intersection_to_z_plane` (normal = +z) */
static o_vec3 line_intersection_to_z_plane(o_vec3 a, o_vec3 b, real z_distance) {
    o_vec3 normal = v3(R(0.0), R(0.0), R(1.0));
    o_vec3 a_to_b = v3_sub(b, a);
    real t = (z_distance - v3_dot(normal, a)) / v3_dot(normal, a_to_b);
    return v3_add(a, v3_scale(a_to_b, t));
}

static inline o_vec3 v3_min(o_vec3 a, o_vec3 b) { return v3(R_MIN(a.x, b.x), R_MIN(a.y, b.y), R_MIN(a.z, b.z)); }
static inline o_vec3 v3_max(o_vec3 a, o_vec3 b) { return v3(R_MAX(a.x, b.x), R_MAX(a.y, b.y), R_MAX(a.z, b.z)); }

/* shader/src/lib.rs:519-580 `write_cluster_data`: one view-space AABB per cluster (x, y, z). */
void o_write_cluster_data(const tr_uniforms* u, const float inverse_perspective[16], const uint32_t screen_dimensions[2],
                          uint32_t num_clusters_z, tr_cluster_aabb* out) {
    real ip[16];
    for (int k = 0; k < 16; ++k) ip[k] = (real)inverse_perspective[k];
    const uint32_t nx = u->num_clusters[0], ny = u->num_clusters[1];
    for (uint32_t z = 0; z < num_clusters_z; ++z)
        for (uint32_t y = 0; y < ny; ++y)
            for (uint32_t x = 0; x < nx; ++x) {
                uint32_t cluster_id = z * nx * ny + y * nx + x;
                real smin[2] = {(real)x * (real)u->cluster_size_in_pixels[0], (real)y * (real)u->cluster_size_in_pixels[1]};
                real smax[2] = {(real)(x + 1u) * (real)u->cluster_size_in_pixels[0],
                                (real)(y + 1u) * (real)u->cluster_size_in_pixels[1]};
                o_vec3 view_space[2];
                for (int k = 0; k < 2; ++k) {
                    const real* p = k ? smax : smin;
                    /* screen_to_clip :540-544 */
                    real px = p[0] / (real)screen_dimensions[0], py = p[1] / (real)screen_dimensions[1];
                    px = px * R(2.0) - R(1.0);
                    py = py * R(2.0) - R(1.0);
                    /* clip_to_view :546-550 */
                    real clip[4] = {px, py, R(0.0), R(1.0)}, v[4];
                    mat4_mul_vec4(ip, clip, v);
                    view_space[k] = v3_div(v3(v[0], v[1], v[2]), v[3]);
                }
                real z_near = o_slice_to_depth(&u->light_clustering_coefficients, z);
                real z_far = o_slice_to_depth(&u->light_clustering_coefficients, z + 1u);
                o_vec3 eye = v3(R(0.0), R(0.0), R(1.0));
                o_vec3 min_near = line_intersection_to_z_plane(eye, view_space[0], z_near);
                o_vec3 min_far = line_intersection_to_z_plane(eye, view_space[0], z_far);
                o_vec3 max_near = line_intersection_to_z_plane(eye, view_space[1], z_near);
                o_vec3 max_far = line_intersection_to_z_plane(eye, view_space[1], z_far);
                o_vec3 mn = v3_min(v3_min(v3_min(min_near, min_far), max_near), max_far);
                o_vec3 mx = v3_max(v3_max(v3_max(min_near, min_far), max_near), max_far);
                tr_cluster_aabb* o = &out[cluster_id];
                memset(o, 0, sizeof(*o));
                o->min[0] = (float)mn.x; o->min[1] = (float)mn.y; o->min[2] = (float)mn.z;
                o->max[0] = (float)mx.x; o->max[1] = (float)mx.y; o->max[2] = (float)mx.z;
            }
}

/* shared-structs/src/lib.rs:290-298 `ClusterAabb::distance_sq` */
static real aabb_distance_sq(const tr_cluster_aabb* c, o_vec3 point) {
    o_vec3 mn = f3(c->min), mx = f3(c->max);
    o_vec3 d = v3_max(v3_max(v3_sub(mn, point), v3_sub(point, mx)), v3_splat(R(0.0)));
    return v3_dot(d, d);
}

/* shared-structs/src/lib.rs:300-319 `ClusterAabb::cull_spotlight` */
static int aabb_cull_spotlight(const tr_cluster_aabb* c, o_vec3 origin, o_vec3 direction, real angle, real range) {
    o_vec3 mn = f3(c->min), mx = f3(c->max);
    o_vec3 center = v3_div(v3_add(mn, mx), R(2.0));
    o_vec3 mc = v3_sub(mx, center);
    real radius = R_SQRT(v3_dot(mc, mc));
    o_vec3 vector = v3_sub(center, origin);
    real vector_len_sq = v3_dot(vector, vector);
    real vector_1_len = v3_dot(vector, direction);
    real vector_1_len_sq = vector_1_len * vector_1_len;
    real distance_closest_point = R_COS(angle) * R_SQRT(vector_len_sq - vector_1_len_sq) - vector_1_len * R_SIN(angle);
    int angle_cull = distance_closest_point > radius;
    int front_cull = vector_1_len > radius + range;
    int back_cull = vector_1_len < -radius;
    return angle_cull || front_cull || back_cull;
}

static inline o_vec3 v3_cross(o_vec3 a, o_vec3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

/* glam 0.19 scalar Quat * Vec3 (q = x, y, z, w) */
static o_vec3 quat_mul_vec3(const real q[4], o_vec3 v) {
    real w = q[3];
    o_vec3 b = v3(q[0], q[1], q[2]);
    real b2 = v3_dot(b, b);
    return v3_add(v3_add(v3_scale(v, w * w - b2), v3_scale(b, v3_dot(v, b) * R(2.0))), v3_scale(v3_cross(b, v), w * R(2.0)));
}

/* shader/src/lib.rs:596-645 `assign_lights_to_clusters`.  The reference appends with atomics in arbitrary
 * order; here lights are visited in ascending order, so every list comes out sorted (deterministic). */
void o_assign_lights_to_clusters(const tr_light* lights, uint32_t num_lights, const tr_cluster_aabb* clusters,
                                 uint32_t num_clusters, const float view_matrix[16], const float view_rotation[4],
                                 uint32_t* counts, uint32_t* indices) {
    real vm[16], q[4];
    for (int k = 0; k < 16; ++k) vm[k] = (real)view_matrix[k];
    for (int k = 0; k < 4; ++k) q[k] = (real)view_rotation[k];
    for (uint32_t c = 0; c < num_clusters; ++c) counts[c] = 0;
    for (uint32_t l = 0; l < num_lights; ++l) {
        const tr_light* light = &lights[l];
        real p4[4] = {light->position_and_spotlight_epsilon[0], light->position_and_spotlight_epsilon[1],
                      light->position_and_spotlight_epsilon[2], R(1.0)}, lp[4];
        mat4_mul_vec4(vm, p4, lp);
        o_vec3 light_position = v3(lp[0], lp[1], lp[2]);
        real falloff_distance_sq = light->colour_emission_and_falloff_distance_sq[3];
        int is_spot = light->spotlight_direction_and_outer_angle[3] != 0.0f;
        o_vec3 spot_dir = v3(R(0.0), R(0.0), R(0.0));
        if (is_spot) spot_dir = quat_mul_vec3(q, f3(light->spotlight_direction_and_outer_angle));
        for (uint32_t c = 0; c < num_clusters; ++c) {
            if (aabb_distance_sq(&clusters[c], light_position) > falloff_distance_sq) continue;
            if (is_spot && aabb_cull_spotlight(&clusters[c], light_position, spot_dir,
                                               light->spotlight_direction_and_outer_angle[3],
                                               falloff_distance_sq /* the reference passes the squared value as range */))
                continue;
            uint32_t off = counts[c]++;
            if (off < TR_MAX_LIGHTS_PER_CLUSTER) indices[(size_t)c * TR_MAX_LIGHTS_PER_CLUSTER + off] = l;
        }
    }
}


/* ------------------------------------------------- frustum culling + demultiplex */

/* `Similarity * Vec3` (shared-structs/src/lib.rs:233-236): translation + (scale * (rotation * vector)) */
o_vec3 o_similarity_mul_vec3(const tr_instance* inst, o_vec3 v) {
    real q[4] = {inst->rotation[0], inst->rotation[1], inst->rotation[2], inst->rotation[3]};
    o_vec3 r = quat_mul_vec3(q, v);
    return v3_add(f3(inst->translation_and_scale), v3_scale(r, (real)inst->translation_and_scale[3]));
}

/* shader/src/lib.rs:438-465 `cull` */
int o_cull(const float packed_bounding_sphere[4], const tr_instance* inst, const tr_culling_push_constants* pc) {
    o_vec3 center = o_similarity_mul_vec3(inst, f3(packed_bounding_sphere));
    real vm[16], c4[4] = {center.x, center.y, center.z, R(1.0)}, v[4];
    for (int k = 0; k < 16; ++k) vm[k] = (real)pc->view[k];
    mat4_mul_vec4(vm, c4, v);
    center = v3(v[0], v[1], -v[2]);              /* "in the view, +z = back so we flip it" */
    real radius = (real)packed_bounding_sphere[3] * (real)inst->translation_and_scale[3];
    int visible = center.z + radius > (real)pc->z_near;
    visible &= center.z * (real)pc->frustum_x_xz[1] - R_ABS(center.x) * (real)pc->frustum_x_xz[0] < radius;
    visible &= center.z * (real)pc->frustum_y_yz[1] - R_ABS(center.y) * (real)pc->frustum_y_yz[0] < radius;
    return !visible;
}

/* shader/src/lib.rs:411-436 */
void o_frustum_culling(const tr_primitive_info* primitives, uint32_t num_primitives, const tr_instance* instances,
                       uint32_t num_instances, const tr_culling_push_constants* pc, uint32_t* instance_counts) {
    for (uint32_t p = 0; p < num_primitives; ++p) instance_counts[p] = 0;   /* src/main.rs:1668-1674 */
    for (uint32_t i = 0; i < num_instances; ++i) {
        const tr_instance* inst = &instances[i];
        if (inst->primitive_id >= num_primitives) continue;                  /* unchecked in the reference */
        if (o_cull(primitives[inst->primitive_id].packed_bounding_sphere, inst, pc)) continue;
        instance_counts[inst->primitive_id] += 1;
    }
}

/* shader/src/lib.rs:469-517; the reference appends with atomics (arbitrary order), here in ascending order */
void o_demultiplex_draws(const tr_primitive_info* primitives, uint32_t num_primitives, const uint32_t* instance_counts,
                         uint32_t draw_counts[4], tr_draw_command* const draws[4]) {
    for (int k = 0; k < 4; ++k) draw_counts[k] = 0;
    for (uint32_t d = 0; d < num_primitives; ++d) {
        uint32_t n = instance_counts[d];
        if (n == 0) continue;
        const tr_primitive_info* p = &primitives[d];
        uint32_t b = p->draw_buffer_index < 3u ? p->draw_buffer_index : 3u;   /* `_ =>` arm */
        tr_draw_command c;
        c.index_count = p->index_count;
        c.instance_count = n;
        c.first_index = p->first_index;
        c.vertex_offset = 0;
        c.first_instance = p->first_instance;
        draws[b][draw_counts[b]++] = c;
    }
}

/* src/main.rs:1726-1746: frustum planes from the perspective matrix rows */
void o_culling_push_constants(const real P[16], const float view_colmajor[16], real z_near, tr_culling_push_constants* out) {
    /* row(i).truncate() of a column-major matrix: (P[i], P[4+i], P[8+i]) */
    o_vec3 r0 = v3(P[0], P[4], P[8]), r1 = v3(P[1], P[5], P[9]), r3 = v3(P[3], P[7], P[11]);
    o_vec3 fx = v3_normalize(v3_add(r3, r0)), fy = v3_normalize(v3_add(r3, r1));
    memset(out, 0, sizeof(*out));
    for (int k = 0; k < 16; ++k) out->view[k] = view_colmajor[k];
    out->frustum_x_xz[0] = (float)fx.x; out->frustum_x_xz[1] = (float)fx.z;
    out->frustum_y_yz[0] = (float)fy.y; out->frustum_y_yz[1] = (float)fy.z;
    out->z_near = (float)z_near;
}


/* ------------------------------------------------- material textures */

/* R8G8B8A8_SRGB texel decode (Khronos data format spec 13.3), exact in `real` */
real o_srgb_to_linear(uint8_t c) {
    real x = (real)c / R(255.0);
    return x <= R(0.04045) ? x / R(12.92) : R_POW((x + R(0.055)) / R(1.055), R(2.4));
}

static uint8_t linear_to_srgb8_tex(real x) { return o_linear_to_srgb8(x); }

void o_texture_layout(uint32_t w, uint32_t h, o_texture* out, uint64_t* total_texels) {
    memset(out, 0, sizeof(*out));
    out->width = w;
    out->height = h;
    out->levels = o_mip_levels_for_size(w, h);   /* src/model_loading.rs:354 */
    if (out->levels > TR_MAX_MIP_LEVELS) out->levels = TR_MAX_MIP_LEVELS;
    uint64_t off = 0;
    for (uint32_t l = 0; l < out->levels; ++l) {
        out->level_offset[l] = (uint32_t)off;
        off += (uint64_t)level_dim(w, l) * level_dim(h, l);
    }
    if (total_texels) *total_texels = off;
}

static void decode_texel(const o_texture* t, const uint8_t* p, real out[4]) {
    for (int k = 0; k < 3; ++k) out[k] = t->srgb ? o_srgb_to_linear(p[k]) : (real)p[k] / R(255.0);
    out[3] = (real)p[3] / R(255.0);
}

/* The mip chain `load_image_from_bytes` builds (un-vendored: restated as the vkCmdBlitImage LINEAR chain, like
 * o_generate_mips); sRGB images are filtered in linear light and re-encoded. UNORM stores round to nearest. */
void o_generate_texture_mips(const o_texture* t, uint8_t* texels) {
    for (uint32_t l = 1; l < t->levels; ++l) {
        uint32_t ws = level_dim(t->width, l - 1), hs = level_dim(t->height, l - 1);
        uint32_t wd = level_dim(t->width, l), hd = level_dim(t->height, l);
        const uint8_t* src = texels + (size_t)t->level_offset[l - 1] * 4u;
        uint8_t* dst = texels + (size_t)t->level_offset[l] * 4u;
        real sx = (real)ws / (real)wd, sy = (real)hs / (real)hd;
        for (uint32_t j = 0; j < hd; ++j) {
            real y = ((real)j + R(0.5)) * sy - R(0.5);
            real fy0 = R_FLOOR(y), by = y - fy0;
            int32_t y0 = (int32_t)fy0, y1 = y0 + 1;
            if (y0 < 0) y0 = 0;
            if (y0 > (int32_t)hs - 1) y0 = (int32_t)hs - 1;
            if (y1 > (int32_t)hs - 1) y1 = (int32_t)hs - 1;
            for (uint32_t i = 0; i < wd; ++i) {
                real x = ((real)i + R(0.5)) * sx - R(0.5);
                real fx0 = R_FLOOR(x), ax = x - fx0;
                int32_t x0 = (int32_t)fx0, x1 = x0 + 1;
                if (x0 < 0) x0 = 0;
                if (x0 > (int32_t)ws - 1) x0 = (int32_t)ws - 1;
                if (x1 > (int32_t)ws - 1) x1 = (int32_t)ws - 1;
                real w00 = (R(1.0) - ax) * (R(1.0) - by), w10 = ax * (R(1.0) - by), w01 = (R(1.0) - ax) * by, w11 = ax * by;
                real t00[4], t10[4], t01[4], t11[4];
                decode_texel(t, src + ((size_t)y0 * ws + (uint32_t)x0) * 4u, t00);
                decode_texel(t, src + ((size_t)y0 * ws + (uint32_t)x1) * 4u, t10);
                decode_texel(t, src + ((size_t)y1 * ws + (uint32_t)x0) * 4u, t01);
                decode_texel(t, src + ((size_t)y1 * ws + (uint32_t)x1) * 4u, t11);
                for (int k = 0; k < 4; ++k) {
                    real r = (t00[k] * w00 + t10[k] * w10) + (t01[k] * w01 + t11[k] * w11);
                    uint8_t b;
                    if (t->srgb && k < 3) b = linear_to_srgb8_tex(r);
                    else {
                        if (!(r > R(0.0))) r = R(0.0);
                        if (r > R(1.0)) r = R(1.0);
                        b = (uint8_t)(r * R(255.0) + R(0.5));
                    }
                    dst[((size_t)j * wd + i) * 4u + k] = b;
                }
            }
        }
    }
}

static void tex_bilinear(const o_texture* t, uint32_t level, real u, real v, real out[4]) {
    uint32_t w = level_dim(t->width, level), h = level_dim(t->height, level);
    const uint8_t* base = t->texels + (size_t)t->level_offset[level] * 4u;
    /* REPEAT (the `sampler` of src/main.rs:683-692 leaves the address mode at its default): wrap the
     * coordinate to [0,1) first, then the usual x = u*w - 0.5 */
    real uu = u - R_FLOOR(u), vv = v - R_FLOOR(v);
    real x = uu * (real)w - R(0.5), y = vv * (real)h - R(0.5);
    real fx0 = R_FLOOR(x), fy0 = R_FLOOR(y);
    real fx = x - fx0, fy = y - fy0;
    int32_t x0 = (int32_t)fx0, y0 = (int32_t)fy0;
    int32_t x1 = x0 + 1, y1 = y0 + 1;
    if (x0 < 0) x0 += (int32_t)w;
    if (y0 < 0) y0 += (int32_t)h;
    if (x1 >= (int32_t)w) x1 -= (int32_t)w;
    if (y1 >= (int32_t)h) y1 -= (int32_t)h;
    if (x0 >= (int32_t)w) x0 = (int32_t)w - 1;   /* only for non-finite coordinates */
    if (y0 >= (int32_t)h) y0 = (int32_t)h - 1;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 < 0 || x1 >= (int32_t)w) x1 = 0;
    if (y1 < 0 || y1 >= (int32_t)h) y1 = 0;
    real t00[4], t10[4], t01[4], t11[4];
    decode_texel(t, base + ((size_t)y0 * w + (uint32_t)x0) * 4u, t00);
    decode_texel(t, base + ((size_t)y0 * w + (uint32_t)x1) * 4u, t10);
    decode_texel(t, base + ((size_t)y1 * w + (uint32_t)x0) * 4u, t01);
    decode_texel(t, base + ((size_t)y1 * w + (uint32_t)x1) * 4u, t11);
    for (int k = 0; k < 4; ++k) {
        real top = t00[k] + (t10[k] - t00[k]) * fx;
        real bot = t01[k] + (t11[k] - t01[k]) * fx;
        out[k] = top + (bot - top) * fy;
    }
}

/* TextureSampler::sample (shader/src/lib.rs:257-261): `texture.sample(sampler, uv)`, implicit LOD.
 * Vulkan 1.3 "Scale Factor Operation / LOD Operation" restated (unpinned, driver territory):
 * rho = max(|(du/dx w, dv/dx h)|, |(du/dy w, dv/dy h)|), lambda = log2(rho), clamped to [0, levels-1],
 * LINEAR mip filter between floor(lambda) and the next level, LINEAR min/mag. */
void o_sample_texture(const o_texture* t, real u, real v, o_vec2 duv_dx, o_vec2 duv_dy, real out[4]) {
    real mxx = duv_dx.x * (real)t->width, mxy = duv_dx.y * (real)t->height;
    real myx = duv_dy.x * (real)t->width, myy = duv_dy.y * (real)t->height;
    real rho_x = R_SQRT(mxx * mxx + mxy * mxy), rho_y = R_SQRT(myx * myx + myy * myy);
    real rho = R_MAX(rho_x, rho_y);
    real lambda = R_LOG2(rho);
    real max_lod = (real)(t->levels - 1u);
    real l = R_MIN(R_MAX(lambda, R(0.0)), max_lod);   /* NaN / -inf -> 0 */
    real lf = R_FLOOR(l), frac = l - lf;
    uint32_t l0 = (uint32_t)lf, l1 = l0 + 1u < t->levels ? l0 + 1u : t->levels - 1u;
    real a[4], b[4];
    tex_bilinear(t, l0, u, v, a);
    tex_bilinear(t, l1, u, v, b);
    for (int k = 0; k < 4; ++k) out[k] = a[k] + (b[k] - a[k]) * frac;
}

static void sample_material_texture(const o_scene* s, int32_t id, o_vec2 uv, const o_frag_derivs* d, real out[4]) {
    o_vec2 zero = {R(0.0), R(0.0)};
    if (id < 0 || (uint32_t)id >= s->num_textures) {   /* unbound slot: robust read */
        out[0] = out[1] = out[2] = out[3] = R(0.0);
        return;
    }
    o_sample_texture(&s->textures[id], uv.x, uv.y, d ? d->duv_dx : zero, d ? d->duv_dy : zero, out);
}

/* shader/src/lighting.rs:243-259 `compute_cotangent_frame` + :222-241 `calculate_normal` */
static o_vec3 calculate_normal(const o_scene* s, o_vec3 interpolated_normal, const tr_material_info* material,
                               o_vec2 uv, const o_frag_derivs* d) {
    o_vec3 normal = v3_normalize(interpolated_normal);
    if (material->textures.normal_map != -1) {
        real smp[4];
        sample_material_texture(s, material->textures.normal_map, uv, d, smp);
        /* `map_normal * 255.0 / 127.0 - 128.0 / 127.0`: the compiled shader (fragment.spv) multiplies by the
         * folded constant 255/127 = 2.007874 and subtracts the folded 128/127 */
        o_vec3 map_normal = v3_add_s(v3_scale(v3(smp[0], smp[1], smp[2]), R(255.0) / R(127.0)), -(R(128.0) / R(127.0)));
        o_vec3 zero3 = v3_splat(R(0.0));
        o_vec2 zero2 = {R(0.0), R(0.0)};
        o_vec3 dp1 = d ? d->dpos_dx : zero3, dp2 = d ? d->dpos_dy : zero3;
        o_vec2 duv1 = d ? d->duv_dx : zero2, duv2 = d ? d->duv_dy : zero2;
        o_vec3 dp2perp = v3_cross(dp2, normal);
        o_vec3 dp1perp = v3_cross(normal, dp1);
        o_vec3 t = v3_add(v3_scale(dp2perp, duv1.x), v3_scale(dp1perp, duv2.x));
        o_vec3 b = v3_add(v3_scale(dp2perp, duv1.y), v3_scale(dp1perp, duv2.y));
        real invmax = R(1.0) / R_SQRT(R_MAX(v3_dot(t, t), v3_dot(b, b)));
        o_vec3 c0 = v3_scale(t, invmax), c1 = v3_scale(b, invmax);
        /* Mat3::from_cols(c0, c1, normal) * map_normal */
        o_vec3 r = v3_add(v3_add(v3_scale(c0, map_normal.x), v3_scale(c1, map_normal.y)), v3_scale(normal, map_normal.z));
        normal = v3_normalize(r);
    }
    return normal;
}

/* shader/src/lighting.rs:261-301 */
static o_material_params get_material_params(const o_scene* s, const real diffuse[4], const tr_material_info* m,
                                             o_vec2 uv, const o_frag_derivs* d) {
    o_material_params mp;
    real metallic = m->metallic_factor, roughness = m->roughness_factor;
    real smp[4];
    if (m->textures.metallic_roughness != -1) {
        sample_material_texture(s, m->textures.metallic_roughness, uv, d, smp);
        metallic *= smp[2];    /* sample.zy: "These two are switched!" */
        roughness *= smp[1];
    }
    o_vec3 specular_colour = f3(m->specular_colour_factor);
    if (m->textures.specular_colour != -1) {
        sample_material_texture(s, m->textures.specular_colour, uv, d, smp);
        specular_colour = v3_mul(specular_colour, v3(smp[0], smp[1], smp[2]));
    }
    real specular_factor = m->specular_factor;
    if (m->textures.specular != -1) {
        sample_material_texture(s, m->textures.specular, uv, d, smp);
        specular_factor *= smp[3];
    }
    mp.diffuse_colour = v3(diffuse[0], diffuse[1], diffuse[2]);
    mp.metallic = metallic;
    mp.perceptual_roughness = roughness;
    mp.index_of_refraction = m->index_of_refraction;
    mp.specular_colour = specular_colour;
    mp.specular_factor = specular_factor;
    return mp;
}

/* shader/src/lighting.rs:303-313 */
static o_vec3 get_emission(const o_scene* s, const tr_material_info* m, o_vec2 uv, const o_frag_derivs* d) {
    o_vec3 emission = f3(m->emissive_factor);
    if (m->textures.emissive != -1) {
        real smp[4];
        sample_material_texture(s, m->textures.emissive, uv, d, smp);
        emission = v3_mul(emission, v3(smp[0], smp[1], smp[2]));
    }
    return emission;
}

/* diffuse = material.diffuse_factor; if textured: diffuse *= sample (lib.rs:65-69 / 190-194) */
static void get_diffuse(const o_scene* s, const tr_material_info* m, o_vec2 uv, const o_frag_derivs* d, real out[4]) {
    for (int k = 0; k < 4; ++k) out[k] = m->diffuse_factor[k];
    if (m->textures.diffuse != -1) {
        real smp[4];
        sample_material_texture(s, m->textures.diffuse, uv, d, smp);
        for (int k = 0; k < 4; ++k) out[k] *= smp[k];
    }
}

/* cluster index: shader/src/lib.rs:88-98 / 205-215 */
static uint32_t cluster_index(const tr_uniforms* u, const real frag_coord[4]) {
    /* (index work: fp32 division in both builds, see o_get_depth_slice) */
    uint32_t cx = f32_as_u32((real)((float)frag_coord[0] / u->cluster_size_in_pixels[0]));
    uint32_t cy = f32_as_u32((real)((float)frag_coord[1] / u->cluster_size_in_pixels[1]));
    uint32_t cz = o_get_depth_slice(&u->light_clustering_coefficients, frag_coord[2]);
    return cz * u->num_clusters[0] * u->num_clusters[1] + cy * u->num_clusters[0] + cx;
}

/* storage-buffer reads out of range behave like robustBufferAccess (read 0): the reference
 * indexes unchecked (shader/src/lib.rs:393-395) and get_depth_slice is not clamped above. */
static uint32_t cluster_count(const o_scene* s, uint32_t cluster) {
    return cluster < s->num_clusters_total ? s->cluster_light_counts[cluster] : 0u;
}

static const o_vec3 DEBUG_COLOURS[15] = { /* shader/src/lib.rs:647-664 */
    {R(0.0), R(0.0), R(0.0)},      {R(0.0), R(0.0), R(0.1647)},   {R(0.0), R(0.0), R(0.3647)}, {R(0.0), R(0.0), R(0.6647)},
    {R(0.0), R(0.0), R(0.9647)},   {R(0.0), R(0.9255), R(0.9255)}, {R(0.0), R(0.5647), R(0.0)}, {R(0.0), R(0.7843), R(0.0)},
    {R(1.0), R(1.0), R(0.0)},      {R(0.90588), R(0.75294), R(0.0)}, {R(1.0), R(0.5647), R(0.0)}, {R(1.0), R(0.0), R(0.0)},
    {R(0.8392), R(0.0), R(0.0)},   {R(1.0), R(0.0), R(1.0)},      {R(0.6), R(0.3333), R(0.7882)}};

/* shader/src/lib.rs:164-249 */
void o_fragment(const o_scene* s, o_vec3 position, o_vec3 normal_in, o_vec2 uv, uint32_t material_id,
                const real frag_coord[4], const o_frag_derivs* d, real out_rgba[4]) {
    const tr_material_info* material = &s->materials[material_id];
    const tr_uniforms* u = &s->uniforms;
    real diffuse[4];
    get_diffuse(s, material, uv, d, diffuse);

    o_vec3 view_vector = v3_sub(f3(s->push.view_position), position);
    o_vec3 view = v3_normalize(view_vector);
    o_vec3 normal = calculate_normal(s, normal_in, material, uv, d);
    o_material_params mp = get_material_params(s, diffuse, material, uv, d);
    o_vec3 emission = get_emission(s, material, uv, d);

    uint32_t cluster = cluster_index(u, frag_coord);
    uint32_t num_lights = cluster_count(s, cluster);

    /* evaluate_lights, lighting.rs:145-220 */
    o_vec3 sun_dir = f3(u->sun_dir);
    o_vec3 sun_intensity = v3_scale(f3(u->sun_intensity), R(1.0));
    o_brdf_result sum = o_basic_brdf(normal, sun_dir, sun_intensity, view, mp);
    uint32_t offset = cluster * TR_MAX_LIGHTS_PER_CLUSTER;
    for (uint32_t cur = offset; cur < offset + num_lights; ++cur) {
        const tr_light* light = &s->lights[s->light_indices[cur]];
        o_vec3 direction;
        real distance, attenuation;
        o_light_direction_and_attenuation(position, f3(light->position_and_spotlight_epsilon),
                                          &direction, &distance, &attenuation);
        real factor = R(1.0);
        if (light->spotlight_direction_and_outer_angle[3] != R(0.0)) factor *= o_spotlight_factor(light, direction);
        o_vec3 light_emission = v3_scale(f3(light->colour_emission_and_falloff_distance_sq), factor);
        o_brdf_result r = o_basic_brdf(normal, direction, v3_scale(light_emission, attenuation), view, mp);
        sum.diffuse = v3_add(sum.diffuse, r.diffuse);
        sum.specular = v3_add(sum.specular, r.specular);
    }

    o_vec3 out = v3_add(v3_add(sum.diffuse, sum.specular), emission);
    if (u->debug_clusters != 0u) { /* :241-245 */
        o_vec3 a = DEBUG_COLOURS[num_lights % 15u];
        o_vec3 b = DEBUG_COLOURS[cluster % 15u];
        out = v3_add(a, v3_scale(v3_add_s(b, -R(0.5)), R(0.025)));
    }
    out_rgba[0] = out.x;
    out_rgba[1] = out.y;
    out_rgba[2] = out.z;
    out_rgba[3] = R(1.0);
}

typedef struct { const o_pyramid* p; } fb_user_t;
typedef struct { const o_scene* s; } lut_user_t;

static o_vec3 fb_sampler_cb(void* user, o_vec2 uv, real lod) {
    return o_sample_pyramid(((fb_user_t*)user)->p, uv.x, uv.y, lod);
}
static o_vec2 lut_sampler_cb(void* user, real nov, real roughness) {
    const o_scene* s = ((lut_user_t*)user)->s;
    return o_sample_lut(s->ggx_lut_rgba8, s->lut_width, s->lut_height, nov, roughness);
}

/* shader/src/lib.rs:37-162 */
void o_fragment_transmission(const o_scene* s, const o_pyramid* framebuffer, o_vec3 position,
                             o_vec3 normal_in, o_vec2 uv, uint32_t material_id, real model_scale,
                             const real frag_coord[4], const o_frag_derivs* d, real out_rgba[4]) {
    const tr_material_info* material = &s->materials[material_id];
    const tr_uniforms* u = &s->uniforms;
    real diffuse[4];
    get_diffuse(s, material, uv, d, diffuse);
    real transmission_factor = material->transmission_factor;
    if (material->textures.transmission != -1) {   /* lib.rs:73-77 */
        real smp[4];
        sample_material_texture(s, material->textures.transmission, uv, d, smp);
        transmission_factor *= smp[0];
    }

    o_vec3 view_vector = v3_sub(f3(s->push.view_position), position);
    o_vec3 view = v3_normalize(view_vector);
    o_vec3 normal = calculate_normal(s, normal_in, material, uv, d);
    o_material_params mp = get_material_params(s, diffuse, material, uv, d);
    o_vec3 emission = get_emission(s, material, uv, d);

    uint32_t cluster = cluster_index(u, frag_coord);
    uint32_t num_lights = cluster_count(s, cluster);

    /* evaluate_lights_transmission, lighting.rs:13-95 (no spotlight factor here) */
    o_vec3 sun_dir = f3(u->sun_dir);
    o_vec3 sun_intensity = v3_scale(f3(u->sun_intensity), R(1.0));
    o_brdf_result sum = o_basic_brdf(normal, sun_dir, sun_intensity, view, mp);
    o_vec3 transmission = v3_mul(sun_intensity, o_transmission_btdf(mp, normal, view, sun_dir));
    uint32_t offset = cluster * TR_MAX_LIGHTS_PER_CLUSTER;
    for (uint32_t cur = offset; cur < offset + num_lights; ++cur) {
        const tr_light* light = &s->lights[s->light_indices[cur]];
        o_vec3 direction;
        real distance, attenuation;
        o_light_direction_and_attenuation(position, f3(light->position_and_spotlight_epsilon),
                                          &direction, &distance, &attenuation);
        o_vec3 light_emission = v3_scale(f3(light->colour_emission_and_falloff_distance_sq), R(1.0));
        o_brdf_result r = o_basic_brdf(normal, direction, v3_scale(light_emission, attenuation), view, mp);
        sum.diffuse = v3_add(sum.diffuse, r.diffuse);
        sum.specular = v3_add(sum.specular, r.specular);
        transmission = v3_add(transmission, v3_mul(v3_scale(light_emission, attenuation),
                                                   o_transmission_btdf(mp, normal, view, direction)));
    }

    real thickness = material->thickness_factor;
    if (material->textures.thickness != -1) {   /* lib.rs:122-124 */
        real smp[4];
        sample_material_texture(s, material->textures.thickness, uv, d, smp);
        thickness *= smp[1];
    }

    o_ibl_volume_refraction_params ip;
    ip.material_params = mp;
    ip.framebuffer_size_x = s->push.framebuffer_size[0];
    ip.normal = normal;
    ip.view = view;
    for (int k = 0; k < 16; ++k) ip.proj_view_matrix[k] = (real)s->push.proj_view[k];
    ip.position = position;
    ip.thickness = thickness;
    ip.model_scale = model_scale;
    ip.attenuation_distance = material->attenuation_distance;
    ip.attenuation_colour = f3(material->attenuation_colour);
    fb_user_t fbu = {framebuffer};
    lut_user_t lu = {s};
    transmission = v3_add(transmission, o_ibl_volume_refraction(&ip, fb_sampler_cb, &fbu, lut_sampler_cb, &lu));

    /* :157-161 (transmission factor applied twice: reference behaviour) */
    o_vec3 real_transmission = v3_scale(transmission, transmission_factor);
    o_vec3 diffuse_out = v3_lerp(sum.diffuse, real_transmission, transmission_factor);
    o_vec3 out = v3_add(v3_add(diffuse_out, sum.specular), emission);
    out_rgba[0] = out.x;
    out_rgba[1] = out.y;
    out_rgba[2] = out.z;
    out_rgba[3] = R(1.0);
}

/* ------------------------------------------------------------ whole passes */

typedef struct {
    const o_scene* s;
    const o_gbuffer* g;
    const o_pyramid* fb;
    tr_rect rect;
    uint32_t y_begin, y_end;
    uint16_t* hdr_f16;
    real* hdr_f32;
    uint16_t* mip0_f16;
    int transmissive;
} band_job;

static void* band_worker(void* arg) {
    band_job* j = (band_job*)arg;
    const o_gbuffer* g = j->g;
    for (uint32_t y = j->y_begin; y < j->y_end; ++y) {
        for (uint32_t x = j->rect.x0; x < j->rect.x1; ++x) {
            /* planes hold the tile at (origin_x, origin_y); colour targets are whole-frame */
            size_t i = (size_t)(y - g->origin_y) * g->width + (x - g->origin_x);
            size_t o = (size_t)y * j->s->push.framebuffer_size[0] + x;
            uint32_t mat = g->material_id[i];
            real rgba[4];
            if (mat == TR_NOT_COVERED) {
                if (j->transmissive) continue; /* attachment LOAD: keep */
                rgba[0] = rgba[1] = rgba[2] = R(0.0); /* clear colour, src/main.rs:1592-1601 */
                rgba[3] = R(1.0);
            } else {
                o_vec3 pos = f3(&g->pos_depth[i * 4]);
                o_vec3 nrm = f3(&g->nrm_scale[i * 4]);
                o_vec2 uv = {g->uv[i * 2], g->uv[i * 2 + 1]};
                real frag_coord[4] = {(real)x + R(0.5), (real)y + R(0.5), g->pos_depth[i * 4 + 3], R(1.0)};
                /* OpDPdx / OpDPdy restated on the G-buffer: differences inside the pixel's 2x2 quad (frame
                 * coordinates), value(x|1) - value(x&~1) and value(y|1) - value(y&~1); zero when the partner is
                 * outside the planes or not covered (a rasteriser would extrapolate a helper invocation). */
                o_frag_derivs dv;
                memset(&dv, 0, sizeof(dv));
                {
                    uint32_t xa = x & ~1u, xb = x | 1u, ya = y & ~1u, yb = y | 1u;
                    uint32_t gx1 = g->origin_x + g->width, gy1 = g->origin_y + g->height;
                    if (xa >= g->origin_x && xb < gx1) {
                        size_t ia = (size_t)(y - g->origin_y) * g->width + (xa - g->origin_x), ib = ia + 1;
                        if (g->material_id[ia] != TR_NOT_COVERED && g->material_id[ib] != TR_NOT_COVERED) {
                            /* the shader differentiates the value -view_vector = -(view_position - position) */
                            o_vec3 eye = f3(j->s->push.view_position);
                            dv.dpos_dx = v3_sub(v3_neg(v3_sub(eye, f3(&g->pos_depth[ib * 4]))),
                                                v3_neg(v3_sub(eye, f3(&g->pos_depth[ia * 4]))));
                            dv.duv_dx.x = (real)g->uv[ib * 2] - (real)g->uv[ia * 2];
                            dv.duv_dx.y = (real)g->uv[ib * 2 + 1] - (real)g->uv[ia * 2 + 1];
                        }
                    }
                    if (ya >= g->origin_y && yb < gy1) {
                        size_t ia = (size_t)(ya - g->origin_y) * g->width + (x - g->origin_x), ib = ia + g->width;
                        if (g->material_id[ia] != TR_NOT_COVERED && g->material_id[ib] != TR_NOT_COVERED) {
                            o_vec3 eye = f3(j->s->push.view_position);
                            dv.dpos_dy = v3_sub(v3_neg(v3_sub(eye, f3(&g->pos_depth[ib * 4]))),
                                                v3_neg(v3_sub(eye, f3(&g->pos_depth[ia * 4]))));
                            dv.duv_dy.x = (real)g->uv[ib * 2] - (real)g->uv[ia * 2];
                            dv.duv_dy.y = (real)g->uv[ib * 2 + 1] - (real)g->uv[ia * 2 + 1];
                        }
                    }
                }
                if (j->transmissive)
                    o_fragment_transmission(j->s, j->fb, pos, nrm, uv, mat, g->nrm_scale[i * 4 + 3], frag_coord, &dv, rgba);
                else
                    o_fragment(j->s, pos, nrm, uv, mat, frag_coord, &dv, rgba);
            }
            for (int k = 0; k < 4; ++k) {
                if (j->hdr_f32) j->hdr_f32[o * 4 + k] = rgba[k];
                if (j->hdr_f16) j->hdr_f16[o * 4 + k] = o_f32_to_f16(rgba[k]);
                if (j->mip0_f16) j->mip0_f16[o * 4 + k] = o_f32_to_f16(rgba[k]);
            }
        }
    }
    return NULL;
}

static void run_bands(band_job proto, int nthreads) {
    uint32_t rows = proto.rect.y1 - proto.rect.y0;
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > rows) nthreads = rows ? (int)rows : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    band_job* jobs = (band_job*)malloc(sizeof(band_job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = proto;
        jobs[t].y_begin = proto.rect.y0 + (uint32_t)(((uint64_t)rows * (uint64_t)t) / (uint64_t)nthreads);
        jobs[t].y_end = proto.rect.y0 + (uint32_t)(((uint64_t)rows * (uint64_t)(t + 1)) / (uint64_t)nthreads);
        if (nthreads == 1) band_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, band_worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

void o_shade_opaque(const o_scene* s, const o_gbuffer* g, tr_rect rect,
                    uint16_t* hdr_f16, real* hdr_f32, uint16_t* opaque_mip0_f16, int nthreads) {
    band_job j;
    memset(&j, 0, sizeof(j));
    j.s = s; j.g = g; j.rect = rect; j.hdr_f16 = hdr_f16; j.hdr_f32 = hdr_f32; j.mip0_f16 = opaque_mip0_f16;
    j.transmissive = 0;
    run_bands(j, nthreads);
}

void o_shade_transmission(const o_scene* s, const o_gbuffer* g, const o_pyramid* framebuffer, tr_rect rect,
                          uint16_t* hdr_f16, real* hdr_f32, int nthreads) {
    band_job j;
    memset(&j, 0, sizeof(j));
    j.s = s; j.g = g; j.fb = framebuffer; j.rect = rect; j.hdr_f16 = hdr_f16; j.hdr_f32 = hdr_f32;
    j.transmissive = 1;
    run_bands(j, nthreads);
}

/* ------------------------------------------------- geometry front end */

/* shader/src/lib.rs:356-385 `vertex_instanced_with_scale` */
void o_vertex_instanced(const tr_instance* inst, const float proj_view[16], o_vec3 position, o_vec3 normal,
                        o_vec3* out_position, o_vec3* out_normal, real out_clip[4], real* out_scale) {
    o_vec3 world = o_similarity_mul_vec3(inst, position);       /* similarity * position */
    if (out_position) *out_position = world;
    if (out_normal) {
        real q[4] = {inst->rotation[0], inst->rotation[1], inst->rotation[2], inst->rotation[3]};
        *out_normal = quat_mul_vec3(q, normal);                  /* similarity.rotation * normal */
    }
    if (out_scale) *out_scale = (real)inst->translation_and_scale[3];
    real pv[16], p4[4] = {world.x, world.y, world.z, R(1.0)};
    for (int k = 0; k < 16; ++k) pv[k] = (real)proj_view[k];
    mat4_mul_vec4(pv, p4, out_clip);                             /* proj_view * position.extend(1.0) */
}

/* shader/src/lib.rs:269-292 `depth_pre_pass_alpha_clip` */
int o_alpha_clip_kills(const o_scene* s, uint32_t material_id, o_vec2 uv, o_vec2 duv_dx, o_vec2 duv_dy) {
    const tr_material_info* m = &s->materials[material_id];
    o_frag_derivs d;
    memset(&d, 0, sizeof(d));
    d.duv_dx = duv_dx;
    d.duv_dy = duv_dy;
    real diffuse[4];
    get_diffuse(s, m, uv, &d, diffuse);
    return diffuse[3] < (real)m->alpha_clipping_cutoff;
}

typedef struct {
    real A[3], B[3], C[3];     /* edge functions, positive inside a front-facing triangle */
    real z[3], w[3];           /* clip z, w per vertex */
    int x0, y0, x1, y1;        /* inclusive pixel bounds, clipped to the frame; empty if x0 > x1 */
    int front;
} o_tri_setup;

static void tri_setup(real clip[3][4], uint32_t width, uint32_t height, o_tri_setup* t) {
    real X[3], Y[3], W[3];
    real hw = R(0.5) * (real)width, hh = R(0.5) * (real)height;
    for (int i = 0; i < 3; ++i) {
        X[i] = (clip[i][0] + clip[i][3]) * hw;     /* pixel x times w */
        Y[i] = (clip[i][1] + clip[i][3]) * hh;
        W[i] = clip[i][3];
        t->z[i] = clip[i][2];
        t->w[i] = clip[i][3];
    }
    for (int i = 0; i < 3; ++i) {
        int j = (i + 1) % 3, k = (i + 2) % 3;
        t->A[i] = Y[k] * W[j] - W[k] * Y[j];        /* (V_k x V_j), V = (X, Y, w) */
        t->B[i] = W[k] * X[j] - X[k] * W[j];
        t->C[i] = X[k] * Y[j] - Y[k] * X[j];
    }
    real det = (X[0] * t->A[0] + Y[0] * t->B[0]) + W[0] * t->C[0];
    t->front = det > R(0.0);                        /* counter-clockwise on screen; also rejects degenerate / NaN */
    t->x0 = 0; t->y0 = 0; t->x1 = (int)width - 1; t->y1 = (int)height - 1;
    if (W[0] > R(0.0) && W[1] > R(0.0) && W[2] > R(0.0)) {
        real xs[3] = {X[0] / W[0], X[1] / W[1], X[2] / W[2]}, ys[3] = {Y[0] / W[0], Y[1] / W[1], Y[2] / W[2]};
        real xmin = R_MIN(xs[0], R_MIN(xs[1], xs[2])), xmax = R_MAX(xs[0], R_MAX(xs[1], xs[2]));
        real ymin = R_MIN(ys[0], R_MIN(ys[1], ys[2])), ymax = R_MAX(ys[0], R_MAX(ys[1], ys[2]));
        /* conservative by one pixel on each side; coverage itself is decided by the edge functions */
        real lim = R(16777216.0);
        xmin = R_MAX(R_MIN(xmin, lim), -lim); xmax = R_MAX(R_MIN(xmax, lim), -lim);
        ymin = R_MAX(R_MIN(ymin, lim), -lim); ymax = R_MAX(R_MIN(ymax, lim), -lim);
        int bx0 = (int)R_FLOOR(xmin) - 1, bx1 = (int)R_FLOOR(xmax) + 1, by0 = (int)R_FLOOR(ymin) - 1, by1 = (int)R_FLOOR(ymax) + 1;
        if (bx0 > t->x0) t->x0 = bx0;
        if (by0 > t->y0) t->y0 = by0;
        if (bx1 < t->x1) t->x1 = bx1;
        if (by1 < t->y1) t->y1 = by1;
    }
}

/* Barycentrics and depth of the triangle at a pixel centre.  Returns 1 when the pixel is covered and inside the
 * clip volume. */
static int tri_pixel(const o_tri_setup* t, real pxc, real pyc, real lambda[3], real* depth) {
    real f[3];
    int inside = 1;
    for (int i = 0; i < 3; ++i) {
        f[i] = (t->A[i] * pxc + t->B[i] * pyc) + t->C[i];
        int tie = t->A[i] > R(0.0) || (t->A[i] == R(0.0) && t->B[i] > R(0.0));
        inside &= f[i] > R(0.0) || (f[i] == R(0.0) && tie);
    }
    real sum = (f[0] + f[1]) + f[2];
    real inv = R(1.0) / sum;
    for (int i = 0; i < 3; ++i) lambda[i] = f[i] * inv;
    real zc = (lambda[0] * t->z[0] + lambda[1] * t->z[1]) + lambda[2] * t->z[2];
    real wc = (lambda[0] * t->w[0] + lambda[1] * t->w[1]) + lambda[2] * t->w[2];
    *depth = zc / wc;
    return inside && sum > R(0.0) && wc > R(0.0) && zc <= wc && *depth > R(0.0);
}

void o_rasterize(const o_scene* s, const o_geometry* geo, const tr_draw_command* const draws[4],
                 const uint32_t draw_counts[4], uint32_t width, uint32_t height, o_layer opaque, o_layer transmissive) {
    size_t npix = (size_t)width * height;
    float* depth0 = (float*)calloc(npix, sizeof(float));
    float* depth1 = (float*)calloc(npix, sizeof(float));
    uint32_t* key = (uint32_t*)calloc(npix, sizeof(uint32_t));
    for (int layer = 0; layer < 2; ++layer) {
        o_layer out = layer ? transmissive : opaque;
        float* depth = layer ? depth1 : depth0;
        for (size_t i = 0; i < npix; ++i) { out.material_id[i] = TR_NOT_COVERED; key[i] = 0; }
        memset(out.pos_depth, 0, npix * 16); memset(out.nrm_scale, 0, npix * 16); memset(out.uv, 0, npix * 8);
        uint32_t t_id = 0;    /* position of the triangle in the layer's draw stream: later wins depth ties */
        for (int b = layer * 2; b < layer * 2 + 2; ++b) {
            for (uint32_t d = 0; d < draw_counts[b]; ++d) {
                const tr_draw_command* dc = &draws[b][d];
                for (uint32_t ii = 0; ii < dc->instance_count; ++ii) {
                    uint32_t inst_id = dc->first_instance + ii;
                    const tr_instance* inst = &geo->instances[inst_id];
                    for (uint32_t tri = 0; tri < dc->index_count / 3u; ++tri, ++t_id) {
                        uint32_t vi[3];
                        o_vec3 wp[3], wn[3];
                        o_vec2 vuv[3];
                        real clip[3][4], scale = R(0.0);
                        for (int k = 0; k < 3; ++k) {
                            vi[k] = geo->index[dc->first_index + tri * 3u + (uint32_t)k] + (uint32_t)dc->vertex_offset;
                            o_vertex_instanced(inst, s->push.proj_view, f3(&geo->position[vi[k] * 3u]), f3(&geo->normal[vi[k] * 3u]),
                                               &wp[k], &wn[k], clip[k], &scale);
                            vuv[k].x = geo->uv[vi[k] * 2u]; vuv[k].y = geo->uv[vi[k] * 2u + 1u];
                        }
                        o_tri_setup ts;
                        tri_setup(clip, width, height, &ts);
                        if (!ts.front) continue;
                        for (int py = ts.y0; py <= ts.y1; ++py) {
                            for (int px = ts.x0; px <= ts.x1; ++px) {
                                real lam[3], dep;
                                if (!tri_pixel(&ts, (real)px + R(0.5), (real)py + R(0.5), lam, &dep)) continue;
                                size_t pi = (size_t)py * width + (size_t)px;
                                float depf = (float)dep;
                                if (layer == 1 && !(depf > depth0[pi])) continue;     /* behind (or at) the opaque surface */
                                if (!(depf > depth[pi] || depf == depth[pi])) continue;   /* GREATER, ties: later drawn wins */
                                o_vec2 uvp = {(lam[0] * vuv[0].x + lam[1] * vuv[1].x) + lam[2] * vuv[2].x,
                                              (lam[0] * vuv[0].y + lam[1] * vuv[1].y) + lam[2] * vuv[2].y};
                                if (b & 1) {   /* alpha clip: implicit-LOD fetch from the quad's uv differences */
                                    o_vec2 quv[2];
                                    for (int a = 0; a < 2; ++a) {   /* partner in x, partner in y */
                                        int qx = a == 0 ? (px ^ 1) : px, qy = a == 1 ? (py ^ 1) : py;
                                        real l2[3], d2;
                                        tri_pixel(&ts, (real)qx + R(0.5), (real)qy + R(0.5), l2, &d2);   /* helper invocation */
                                        quv[a].x = (l2[0] * vuv[0].x + l2[1] * vuv[1].x) + l2[2] * vuv[2].x;
                                        quv[a].y = (l2[0] * vuv[0].y + l2[1] * vuv[1].y) + l2[2] * vuv[2].y;
                                    }
                                    real sx = (px & 1) ? R(-1.0) : R(1.0), sy = (py & 1) ? R(-1.0) : R(1.0);
                                    o_vec2 ddx = {(quv[0].x - uvp.x) * sx, (quv[0].y - uvp.y) * sx};
                                    o_vec2 ddy = {(quv[1].x - uvp.x) * sy, (quv[1].y - uvp.y) * sy};
                                    if (o_alpha_clip_kills(s, inst->material_id, uvp, ddx, ddy)) continue;
                                }
                                depth[pi] = depf;
                                key[pi] = t_id;
                                out.material_id[pi] = inst->material_id;
                                for (int c = 0; c < 3; ++c) {
                                    const real* P0 = &wp[0].x; const real* P1 = &wp[1].x; const real* P2 = &wp[2].x;
                                    const real* N0 = &wn[0].x; const real* N1 = &wn[1].x; const real* N2 = &wn[2].x;
                                    out.pos_depth[pi * 4 + (size_t)c] = (float)((lam[0] * P0[c] + lam[1] * P1[c]) + lam[2] * P2[c]);
                                    out.nrm_scale[pi * 4 + (size_t)c] = (float)((lam[0] * N0[c] + lam[1] * N1[c]) + lam[2] * N2[c]);
                                }
                                out.pos_depth[pi * 4 + 3] = depf;
                                out.nrm_scale[pi * 4 + 3] = (float)scale;
                                out.uv[pi * 2] = (float)uvp.x;
                                out.uv[pi * 2 + 1] = (float)uvp.y;
                            }
                        }
                    }
                }
            }
        }
    }
    free(depth0); free(depth1); free(key);
}


/* ------------------------------------------------------------- tonemap */

/* shader/src/tonemapping.rs:8-27 `LottesTonemapper::tonemap` */
void o_lottes_tonemap(const real color[3], const tr_tonemap_params* p, real out[3]) {
    o_vec3 c = v3(color[0], color[1], color[2]);
    real mx = v3_max_element(c);
    o_vec3 ratio = v3_div(c, mx);
    real z = R_POW(mx, (real)p->a);                                   /* tonemap_inner :9-12 */
    real tonemapped_max = z / (R_POW(z, (real)p->d) * (real)p->b + (real)p->c);
    real e1 = (real)p->saturation / (real)p->cross_saturation;
    ratio = v3(R_POW(ratio.x, e1), R_POW(ratio.y, e1), R_POW(ratio.z, e1));
    ratio = v3_lerp(ratio, v3_splat(R(1.0)), R_POW(tonemapped_max, (real)p->crosstalk));
    real e2 = (real)p->cross_saturation;
    ratio = v3(R_POW(ratio.x, e2), R_POW(ratio.y, e2), R_POW(ratio.z, e2));
    o_vec3 r = v3_scale(ratio, tonemapped_max);
    r = v3_max(v3_min(r, v3_splat(R(1.0))), v3_splat(R(0.0)));        /* .min(ONE).max(ZERO); f32::min/max prefer the number */
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* The sRGB encode of the B8G8R8A8_SRGB swapchain (src/main.rs:175): fixed function, restated from the
 * Vulkan/Khronos data format specification, round to nearest. */
uint8_t o_linear_to_srgb8(real x) {
    if (!(x > R(0.0))) x = R(0.0);
    if (x > R(1.0)) x = R(1.0);
    real e = x <= R(0.0031308) ? R(12.92) * x : R(1.055) * R_POW(x, R(1.0) / R(2.4)) - R(0.055);
    return (uint8_t)(e * R(255.0) + R(0.5));
}

/* fragment_tonemap over a whole RGBA16F frame -> RGBA8 sRGB (alpha 255) */
void o_tonemap_frame(const uint16_t* hdr, uint32_t n, const tr_tonemap_params* p, uint8_t* out_rgba8, real* out_linear) {
    for (uint32_t i = 0; i < n; ++i) {
        real c[3] = {o_f16_to_f32(hdr[i * 4]), o_f16_to_f32(hdr[i * 4 + 1]), o_f16_to_f32(hdr[i * 4 + 2])}, t[3];
        o_lottes_tonemap(c, p, t);
        for (int k = 0; k < 3; ++k) {
            if (out_linear) out_linear[i * 3 + k] = t[k];
            if (out_rgba8) out_rgba8[i * 4 + k] = o_linear_to_srgb8(t[k]);
        }
        if (out_rgba8) out_rgba8[i * 4 + 3] = 255;
    }
}

/* ------------------------------------------------------------------ batch forms of the glam-pbr API */
static o_material_params mp_of(const tr_material_params* m) {
    o_material_params o;
    o.diffuse_colour = f3(m->diffuse_colour);
    o.metallic = m->metallic;
    o.perceptual_roughness = m->perceptual_roughness;
    o.index_of_refraction = m->index_of_refraction;
    o.specular_colour = f3(m->specular_colour);
    o.specular_factor = m->specular_factor;
    return o;
}
static void put3(double* out, o_vec3 v) { out[0] = v.x; out[1] = v.y; out[2] = v.z; }

void o_basic_brdf_batch(const tr_basic_brdf_params* p, uint32_t n, double* out) {
    for (uint32_t i = 0; i < n; ++i) {
        o_brdf_result r = o_basic_brdf(f3(p[i].normal), f3(p[i].light), f3(p[i].light_intensity), f3(p[i].view),
                                       mp_of(&p[i].material_params));
        put3(out + 6 * (size_t)i, r.diffuse);
        put3(out + 6 * (size_t)i + 3, r.specular);
    }
}

void o_transmission_btdf_batch(const tr_transmission_btdf_params* p, uint32_t n, double* out) {
    for (uint32_t i = 0; i < n; ++i)
        put3(out + 3 * (size_t)i, o_transmission_btdf(mp_of(&p[i].material_params), f3(p[i].normal), f3(p[i].view), f3(p[i].light)));
}

typedef struct { const uint8_t* rgba8; uint32_t w, h; } lut_image_t;
static o_vec2 lut_image_cb(void* user, real nov, real roughness) {
    const lut_image_t* l = (const lut_image_t*)user;
    return o_sample_lut(l->rgba8, l->w, l->h, nov, roughness);
}

void o_ibl_volume_refraction_batch(const tr_ibl_volume_refraction_params* p, uint32_t n, const o_pyramid* framebuffer,
                                   const uint8_t* ggx_lut_rgba8, uint32_t lut_width, uint32_t lut_height, double* out) {
    fb_user_t fbu = {framebuffer};
    lut_image_t lu = {ggx_lut_rgba8, lut_width, lut_height};
    for (uint32_t i = 0; i < n; ++i) {
        o_ibl_volume_refraction_params q;
        q.material_params = mp_of(&p[i].material_params);
        q.framebuffer_size_x = p[i].framebuffer_size_x;
        q.normal = f3(p[i].normal);
        q.view = f3(p[i].view);
        for (int k = 0; k < 16; ++k) q.proj_view_matrix[k] = p[i].proj_view_matrix[k];
        q.position = f3(p[i].position);
        q.thickness = p[i].thickness;
        q.model_scale = p[i].model_scale;
        q.attenuation_distance = p[i].attenuation_distance;
        q.attenuation_colour = f3(p[i].attenuation_colour);
        put3(out + 3 * (size_t)i, o_ibl_volume_refraction(&q, fb_sampler_cb, &fbu, lut_image_cb, &lu));
    }
}

void o_light_direction_and_attenuation_batch(const float* fp, const float* lp, uint32_t n, double* out) {
    for (uint32_t i = 0; i < n; ++i) {
        o_vec3 d;
        real dist, att;
        o_light_direction_and_attenuation(f3(fp + 3 * (size_t)i), f3(lp + 3 * (size_t)i), &d, &dist, &att);
        put3(out + 5 * (size_t)i, d);
        out[5 * (size_t)i + 3] = dist;
        out[5 * (size_t)i + 4] = att;
    }
}

void o_d_ggx_batch(const float* noh, const float* roughness, uint32_t n, double* out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = o_d_ggx(noh[i], roughness[i]);
}

void o_v_smith_ggx_correlated_batch(const float* nov, const float* nol, const float* roughness, uint32_t n, double* out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = o_v_smith_ggx_correlated(nov[i], nol[i], roughness[i]);
}

void o_fresnel_schlick_batch(const float* voh, const float* f0, const float* f90, uint32_t n, double* out) {
    for (uint32_t i = 0; i < n; ++i)
        put3(out + 3 * (size_t)i, o_fresnel_schlick(voh[i], f3(f0 + 3 * (size_t)i), f3(f90 + 3 * (size_t)i)));
}

void o_compute_f0_batch(const float* metallic, const float* ior, const float* diffuse, uint32_t n, double* out) {
    for (uint32_t i = 0; i < n; ++i) put3(out + 3 * (size_t)i, o_compute_f0(metallic[i], ior[i], f3(diffuse + 3 * (size_t)i)));
}
