/*
 * tr_oracle.h — CPU restatement (plain C, scalar IEEE fp32) of the reference's
 * opaque -> mip chain -> transmissive shading path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it, and
 * only as the checker / the reported CPU baseline.  The product (libtr_shade.so) never
 * links, loads or calls it.
 *
 * Pinning: the reference (expenses/transmission-renderer) ships no tests, golden
 * vectors or fixtures, and neither Rust nor Vulkan exist in the build image, so the
 * Rust sources cannot be run.  What CAN be run are the reference's own committed
 * SPIR-V binaries: oracle/spirv_ref/ interprets the .spv files of compiled-shaders/normal
 * word by word (fixed-function sampling restated, see there) and the fixtures under
 * tests/golden/ were produced that way.  tests/test_oracle_vs_spirv.py pins every
 * entry point of this file against those fixtures.  Steps with no in-tree definition
 * (mip blit filter, texel filtering weights, real->half rounding) are restated from
 * the Vulkan specification and are called out as "unpinned" where they occur.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).  Build: see oracle/Makefile (-ffp-contract=off, no fast-math).
 */
#ifndef TR_ORACLE_H
#define TR_ORACLE_H

#include <stdint.h>
#include "../include/tr_shade.h" /* wire-struct layouts only */

/* `real` is float in libtr_oracle.so (the reference's fp32 semantics: THE oracle) and double in
 * libtr_oracle64.so (same formulas in fp64: used only to measure how much of a pixel's value is
 * fp32 rounding noise of the reference's own formulas, i.e. how ill-conditioned it is). */
#if defined(O_REAL_DOUBLE)
typedef double real;
#else
typedef float real;
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { real x, y, z; } o_vec3;
typedef struct { real x, y; } o_vec2;

/* glam-pbr/src/lib.rs:163-171 `MaterialParams` */
typedef struct {
    o_vec3 diffuse_colour;
    real  metallic;
    real  perceptual_roughness;
    real  index_of_refraction;
    o_vec3 specular_colour;
    real  specular_factor;
} o_material_params;

/* glam-pbr/src/lib.rs:437-441 `BrdfResult` */
typedef struct { o_vec3 diffuse, specular; } o_brdf_result;

/* glam-pbr/src/lib.rs:235-246 `IblVolumeRefractionParams` */
typedef struct {
    o_material_params material_params;
    uint32_t framebuffer_size_x;
    o_vec3 normal;
    o_vec3 view;
    real  proj_view_matrix[16]; /* column-major */
    o_vec3 position;
    real  thickness;
    real  model_scale;
    real  attenuation_distance;
    o_vec3 attenuation_colour;
} o_ibl_volume_refraction_params;

typedef o_vec3 (*o_framebuffer_sampler)(void* user, o_vec2 uv, real lod);
typedef o_vec2 (*o_ggx_lut_sampler)(void* user, real normal_dot_view, real perceptual_roughness);

/* ---- glam-pbr public API (glam-pbr/src/lib.rs) ---- */
void   o_light_direction_and_attenuation(o_vec3 fragment_position, o_vec3 light_position,
                                         o_vec3* direction, real* distance, real* attenuation);
real  o_dot_clamped(o_vec3 a, o_vec3 b);                         /* Dot::new :93-98 */
real  o_d_ggx(real noh, real actual_roughness);                /* :101-109 */
real  o_v_smith_ggx_correlated(real nov, real nol, real actual_roughness); /* :114-133 */
o_vec3 o_fresnel_schlick(real voh, o_vec3 f0, o_vec3 f90);       /* :137-139 */
real  o_to_dielectric_f0(real ior);                             /* :190-195 */
o_vec3 o_transmission_btdf(o_material_params m, o_vec3 normal, o_vec3 view, o_vec3 light); /* :200-233 */
o_vec3 o_refract(o_vec3 incident, o_vec3 normal, real ior);      /* :248-256 */
o_vec3 o_apply_volume_attenuation(o_vec3 transmitted_light, real transmission_distance,
                                  real attenuation_distance, o_vec3 attenuation_colour); /* :275-290 */
o_vec3 o_ibl_volume_refraction(const o_ibl_volume_refraction_params* p,
                               o_framebuffer_sampler fb, void* fb_user,
                               o_ggx_lut_sampler lut, void* lut_user);     /* :292-354 */
o_brdf_result o_basic_brdf(o_vec3 normal, o_vec3 light, o_vec3 light_intensity, o_vec3 view,
                           o_material_params m);                  /* :377-423 */
o_vec3 o_compute_f0(real metallic, real ior, o_vec3 diffuse_colour); /* :454-465 */

/* ---- shared-structs helpers ---- */
void     o_light_cluster_coefficients_new(real z_near, real z_far, uint32_t slices,
                                          tr_light_cluster_coefficients* out);  /* shared-structs:44-52 */
uint32_t o_get_depth_slice(const tr_light_cluster_coefficients* c, real frag_depth); /* :54-63 */
real    o_spotlight_factor(const tr_light* l, o_vec3 direction_to_light);      /* :129-138 */

/* ---- clustered-light build ---- */
real o_slice_to_depth(const tr_light_cluster_coefficients* c, uint32_t slice);  /* shared-structs:65-67 */
void o_write_cluster_data(const tr_uniforms* u, const float inverse_perspective[16], const uint32_t screen_dimensions[2],
                          uint32_t num_clusters_z, tr_cluster_aabb* out);       /* shader/src/lib.rs:519-580 */
void o_assign_lights_to_clusters(const tr_light* lights, uint32_t num_lights, const tr_cluster_aabb* clusters,
                                 uint32_t num_clusters, const float view_matrix[16], const float view_rotation[4],
                                 uint32_t* counts, uint32_t* indices);          /* shader/src/lib.rs:596-645, sorted lists */

/* ---- frustum culling + draw demultiplex ---- */
o_vec3 o_similarity_mul_vec3(const tr_instance* inst, o_vec3 v);               /* shared-structs:233-236 */
int  o_cull(const float packed_bounding_sphere[4], const tr_instance* inst, const tr_culling_push_constants* pc); /* shader/src/lib.rs:438-465 */
void o_frustum_culling(const tr_primitive_info* primitives, uint32_t num_primitives, const tr_instance* instances,
                       uint32_t num_instances, const tr_culling_push_constants* pc, uint32_t* instance_counts); /* :411-436 */
void o_demultiplex_draws(const tr_primitive_info* primitives, uint32_t num_primitives, const uint32_t* instance_counts,
                         uint32_t draw_counts[4], tr_draw_command* const draws[4]);   /* :469-517, ascending order */
void o_culling_push_constants(const real perspective_colmajor[16], const float view_colmajor[16], real z_near,
                              tr_culling_push_constants* out);                        /* src/main.rs:1728-1746 */

/* ---- tonemap ---- */
void    o_lottes_tonemap(const real color[3], const tr_tonemap_params* p, real out[3]);  /* shader/src/tonemapping.rs:8-27 */
uint8_t o_linear_to_srgb8(real x);
void    o_tonemap_frame(const uint16_t* hdr, uint32_t n, const tr_tonemap_params* p, uint8_t* out_rgba8, real* out_linear);

/* ---- host helpers on the path ---- */
uint32_t o_mip_levels_for_size(uint32_t w, uint32_t h);           /* src/main.rs:2590-2592 */
void     o_perspective_matrix_reversed(uint32_t w, uint32_t h, real out_colmajor[16]); /* src/main.rs:39-54 */
void     o_sun_as_normal(real pitch, real yaw, real out[3]);   /* src/main.rs:2715-2722 */

/* ---- half conversion (RTNE; rounding mode is unpinned in Vulkan) ---- */
uint16_t o_f32_to_f16(real f);
real    o_f16_to_f32(uint16_t h);

/* ---- fixed-function sampling restated from the Vulkan spec (unpinned) ---- */
typedef struct {
    const uint16_t* texels;   /* RGBA16F, levels packed as tr_pyramid */
    uint32_t width, height, levels;
    uint32_t level_offset[TR_MAX_MIP_LEVELS];
} o_pyramid;

void   o_pyramid_layout(uint32_t w, uint32_t h, o_pyramid* out, uint64_t* total_texels);
o_vec3 o_sample_pyramid(const o_pyramid* p, real u, real v, real lod);  /* clamp_sampler, trilinear */
o_vec2 o_sample_lut(const uint8_t* rgba8, uint32_t w, uint32_t h, real u, real v); /* bilinear, clamp */
void   o_generate_mips(const o_pyramid* p, uint16_t* texels);      /* generate_mips, src/main.rs:2054 */

/* ---- material textures: RGBA8, full chain of mip_levels_for_size levels packed level
 *      after level; `srgb` selects R8G8B8A8_SRGB decoding (src/model_loading.rs:347-351). ---- */
typedef struct {
    const uint8_t* texels;
    uint32_t width, height, levels, srgb;
    uint32_t level_offset[TR_MAX_MIP_LEVELS];   /* in texels */
} o_texture;

/* screen-space derivatives of the interpolants a fragment shader may differentiate (OpDPdx / OpDPdy) */
typedef struct {
    o_vec3 dpos_dx, dpos_dy;   /* of the value the shader differentiates: -view_vector = -(view_position - position)
                                * (lighting.rs:237), i.e. neighbour's value minus this quad partner's value */
    o_vec2 duv_dx, duv_dy;
} o_frag_derivs;

void   o_texture_layout(uint32_t w, uint32_t h, o_texture* out, uint64_t* total_texels);
void   o_generate_texture_mips(const o_texture* t, uint8_t* texels);   /* LINEAR blit chain, sRGB-aware */
void   o_sample_texture(const o_texture* t, real u, real v, o_vec2 duv_dx, o_vec2 duv_dy, real out_rgba[4]);
real   o_srgb_to_linear(uint8_t c);

/* ---- scene tables shared by both fragment entry points ---- */
typedef struct {
    const tr_material_info* materials;  uint32_t num_materials;
    const tr_light* lights;             uint32_t num_lights;
    const uint32_t* cluster_light_counts;
    const uint32_t* light_indices;
    uint32_t num_clusters_total;
    const uint8_t* ggx_lut_rgba8; uint32_t lut_width, lut_height;
    tr_uniforms uniforms;
    tr_push_constants push;
    const o_texture* textures; uint32_t num_textures;   /* bindless `textures[]` (set 0 binding 0) */
} o_scene;

/* shader/src/lib.rs:164-249 `fragment`.  `d` (may be NULL = zero derivatives) feeds implicit-LOD texture
 * fetches and the normal-map cotangent frame; untextured materials never look at it. */
void o_fragment(const o_scene* s, o_vec3 position, o_vec3 normal, o_vec2 uv, uint32_t material_id,
                const real frag_coord[4], const o_frag_derivs* d, real out_rgba[4]);
/* shader/src/lib.rs:37-162 `fragment_transmission`. */
void o_fragment_transmission(const o_scene* s, const o_pyramid* framebuffer, o_vec3 position,
                             o_vec3 normal, o_vec2 uv, uint32_t material_id, real model_scale,
                             const real frag_coord[4], const o_frag_derivs* d, real out_rgba[4]);

/* ---- whole passes over TGB-v1 planes (host memory). out_f32 (optional) receives the
 *      un-rounded fp32 RGBA; out_f16 (optional) the RTNE RGBA16F target. ---- */
typedef struct {
    const float* pos_depth; const float* nrm_scale; const float* uv; const uint32_t* material_id;
    uint32_t width, height;      /* plane size */
    uint32_t origin_x, origin_y; /* frame position of plane element (0,0), as tr_gbuffer */
} o_gbuffer;

void o_shade_opaque(const o_scene* s, const o_gbuffer* g, tr_rect rect,
                    uint16_t* hdr_f16, real* hdr_f32, uint16_t* opaque_mip0_f16, int nthreads);
void o_shade_transmission(const o_scene* s, const o_gbuffer* g, const o_pyramid* framebuffer, tr_rect rect,
                          uint16_t* hdr_f16, real* hdr_f32, int nthreads);

/* ---- geometry front end: vertex stage + rasterisation into the two TGB-v1 layers ---- */
typedef struct {
    const float* position;   /* 3 floats / vertex (binding 0, stride 12; src/pipelines.rs:287-298) */
    const float* normal;     /* 3 floats / vertex */
    const float* uv;         /* 2 floats / vertex */
    const uint32_t* index;   /* UINT32 triangle list (src/main.rs:1893-1898) */
    const tr_instance* instances;
    uint32_t num_vertices, num_indices, num_instances;
} o_geometry;

typedef struct {             /* one writable TGB-v1 layer, whole frame */
    float* pos_depth; float* nrm_scale; float* uv; uint32_t* material_id;
} o_layer;

/* vertex_instanced_with_scale (shader/src/lib.rs:356-385); vertex_instanced (:330-354), depth_pre_pass_instanced
 * (:316-328) and depth_pre_pass_vertex_alpha_clip (:294-314) compute subsets of the same values. */
void o_vertex_instanced(const tr_instance* inst, const float proj_view[16], o_vec3 position, o_vec3 normal,
                        o_vec3* out_position, o_vec3* out_normal, real out_clip[4], real* out_scale);
/* depth_pre_pass_alpha_clip (shader/src/lib.rs:269-292): 1 = the fragment is killed */
int  o_alpha_clip_kills(const o_scene* s, uint32_t material_id, o_vec2 uv, o_vec2 duv_dx, o_vec2 duv_dy);
/*
 * The fixed-function part between the vertex stage and the fragment entry points, restated (unpinned: Vulkan
 * rasterisation rules, no reference source): back-face culling (front = counter-clockwise, glTF convention),
 * clip-space (homogeneous) edge functions evaluated at pixel centres with a top-left style tie rule, clip volume
 * 0 <= z <= w, perspective-correct interpolation, depth = z/w with GREATER against a buffer cleared to 0
 * (reversed-Z; src/main.rs:1585-1591, src/pipelines.rs:350-398); among equal depths the later-drawn fragment wins.
 * Layer 0 <- draw buffers 0 (opaque) and 1 (alpha clip); layer 1 <- buffers 2 and 3, kept only where nearer than
 * layer 0 (the transmissive depth pre-pass runs against the opaque depth; src/main.rs:2005-2042).
 * `s` supplies materials / textures for the alpha-clip kill and push.proj_view.
 */
void o_rasterize(const o_scene* s, const o_geometry* geo, const tr_draw_command* const draws[4],
                 const uint32_t draw_counts[4], uint32_t width, uint32_t height, o_layer opaque, o_layer transmissive);

/* ---- batch forms of the glam-pbr API: the checker of tr_basic_brdf & co (include/tr_shade.h).  Float records in,
 * `double` out, so libtr_oracle.so (fp32 arithmetic) and libtr_oracle64.so (fp64) share the signatures. ---- */
void o_basic_brdf_batch(const tr_basic_brdf_params* p, uint32_t n, double* out6);
void o_transmission_btdf_batch(const tr_transmission_btdf_params* p, uint32_t n, double* out3);
void o_ibl_volume_refraction_batch(const tr_ibl_volume_refraction_params* p, uint32_t n, const o_pyramid* framebuffer,
                                   const uint8_t* ggx_lut_rgba8, uint32_t lut_width, uint32_t lut_height, double* out3);
void o_light_direction_and_attenuation_batch(const float* fragment_position3, const float* light_position3, uint32_t n,
                                             double* out5);
void o_d_ggx_batch(const float* noh, const float* roughness, uint32_t n, double* out);
void o_v_smith_ggx_correlated_batch(const float* nov, const float* nol, const float* roughness, uint32_t n, double* out);
void o_fresnel_schlick_batch(const float* voh, const float* f0_3, const float* f90_3, uint32_t n, double* out3);
void o_compute_f0_batch(const float* metallic, const float* ior, const float* diffuse3, uint32_t n, double* out3);

#ifdef __cplusplus
}
#endif
#endif
