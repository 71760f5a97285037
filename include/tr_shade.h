/*
 * tr_shade.h — C ABI of the MI355X-native transmission / volume PBR shading path.
 *
 * This is the drop-in boundary for the hot path of expenses/transmission-renderer:
 *   opaque colour pass  ->  opaque-framebuffer mip chain  ->  transmissive pass
 * (reference: src/main.rs:1969-2124 schedules it, shader/src/lib.rs:37-249 are the
 * per-pixel entry points, glam-pbr/src/lib.rs is the math).  The reference has no FFI
 * of its own: its boundary is the Vulkan descriptor/push-constant ABI of the two
 * fragment entry points.  Every struct below mirrors one of those wire structs byte
 * for byte (offsets verified against the OpMemberDecorate Offset words of
 * compiled-shaders/normal/fragment_transmission.spv), and every function replaces one
 * step of `record()`; the reference interface each one replaces is cited.
 *
 * Conventions
 *   - plain C, no C++/torch types; device pointers are raw `void*` owned by the caller
 *   - every launch takes a `hipStream_t` passed as `void*` and is asynchronous
 *   - every function returns a tr_status (0 = OK); no exceptions cross the boundary
 *   - one host thread per context; no global state beyond the opaque tr_context
 *   - there is NO CPU fallback: without a HIP device the context cannot be created
 */
#ifndef TR_SHADE_H
#define TR_SHADE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TR_ABI_VERSION 1u

/* ------------------------------------------------------------------ status */
typedef int32_t tr_status;
enum {
    TR_OK = 0,
    TR_ERR_INVALID_ARGUMENT = 1,
    TR_ERR_NO_DEVICE = 2,        /* no HIP device / runtime error at context creation */
    TR_ERR_HIP = 3,              /* a HIP runtime call failed; see tr_last_hip_error */
    TR_ERR_TABLES_MISSING = 4,   /* a pass was launched before its tables were set  */
    TR_ERR_OUT_OF_MEMORY = 5,
    TR_ERR_UNSUPPORTED = 6,
    TR_ERR_COMM = 7              /* RCCL could not be loaded or a collective call failed; see tr_comm_last_error */
};

/* ------------------------------------------------- wire structs (reference) */

/* shared-structs/src/lib.rs:8-16 `PushConstants` (96 B). proj_view is column-major. */
typedef struct tr_push_constants {
    float    proj_view[16];                    /* @0  */
    float    view_position[3];                 /* @64 (Vec3A: 16-byte slot) */
    float    _pad0;
    uint32_t framebuffer_size[2];              /* @80 */
    uint64_t acceleration_structure_address;   /* @88 (unused: no ray tracing on CDNA) */
} tr_push_constants;

/* shared-structs/src/lib.rs:31-41 `LightClusterCoefficients` (20 B, padded to 32 in Uniforms). */
typedef struct tr_light_cluster_coefficients {
    float    z_near;            /* @0  */
    float    z_far;             /* @4  */
    float    scale;             /* @8  */
    float    bias;              /* @12 */
    uint32_t num_depth_slices;  /* @16 */
    uint32_t _pad[3];
} tr_light_cluster_coefficients;

/* shared-structs/src/lib.rs:18-29 `Uniforms` (96 B). */
typedef struct tr_uniforms {
    tr_light_cluster_coefficients light_clustering_coefficients; /* @0  */
    float    sun_dir[3];                /* @32 */
    float    _pad0;
    float    sun_intensity[3];          /* @48 */
    float    _pad1;
    float    cluster_size_in_pixels[2]; /* @64 */
    uint32_t num_clusters[2];           /* @72 */
    uint32_t debug_clusters;            /* @80 */
    uint32_t ggx_lut_texture_index;     /* @84 (kept for layout; the LUT is bound via tr_upload_ggx_lut) */
    uint32_t _pad2[2];
} tr_uniforms;

/* shared-structs/src/lib.rs:141-153 `Textures` (9 x i32, -1 = none). */
typedef struct tr_textures {
    int32_t diffuse;
    int32_t metallic_roughness;
    int32_t normal_map;
    int32_t emissive;
    int32_t occlusion;
    int32_t transmission;
    int32_t thickness;
    int32_t specular;
    int32_t specular_colour;
} tr_textures;

/* shared-structs/src/lib.rs:155-173 `MaterialInfo` (stride 160 B). */
typedef struct tr_material_info {
    tr_textures textures;            /* @0   */
    float metallic_factor;           /* @36  */
    float roughness_factor;          /* @40  */
    float alpha_clipping_cutoff;     /* @44  */
    float diffuse_factor[4];         /* @48  */
    float emissive_factor[3];        /* @64  */
    float _pad0;
    float normal_map_scale;          /* @80  (never read by the reference shaders) */
    float occlusion_strength;        /* @84  (never read by the reference shaders) */
    float index_of_refraction;       /* @88  */
    float transmission_factor;       /* @92  */
    float thickness_factor;          /* @96  */
    float attenuation_distance;      /* @100 (+INF = no attenuation) */
    float _pad1[2];
    float attenuation_colour[3];     /* @112 */
    float _pad2;
    float specular_factor;           /* @128 */
    float _pad3[3];
    float specular_colour_factor[3]; /* @144 */
    float _pad4;
} tr_material_info;

/* shared-structs/src/lib.rs:70-78 `Light` (stride 48 B). */
typedef struct tr_light {
    float position_and_spotlight_epsilon[4];          /* @0  */
    float colour_emission_and_falloff_distance_sq[4]; /* @16 */
    float spotlight_direction_and_outer_angle[4];     /* @32 (w == 0 => point light) */
} tr_light;

/* shared-structs/src/lib.rs:282-288 `ClusterAabb` (32 B, view space). */
typedef struct tr_cluster_aabb {
    float min[3];
    float _pad0;
    float max[3];
    float _pad1;
} tr_cluster_aabb;

/* ---- geometry and draw records (GPU culling architecture, shared-structs/src/lib.rs:238-281) ---- */

/* `Instance` (stride 48 B): PackedSimilarity{translation.xyz, scale; rotation quaternion xyzw}, ids. */
typedef struct tr_instance {
    float translation_and_scale[4];  /* @0  */
    float rotation[4];               /* @16 (x, y, z, w) */
    uint32_t primitive_id;           /* @32 */
    uint32_t material_id;            /* @36 */
    uint32_t _pad[2];
} tr_instance;

/* `PrimitiveInfo` (stride 32 B): one per drawable primitive; doubles as the draw description. */
typedef struct tr_primitive_info {
    float packed_bounding_sphere[4]; /* @0  centre xyz (model space), radius */
    uint32_t draw_buffer_index;      /* @16 0 opaque, 1 alpha clip, 2 transmission, 3 transmission + alpha clip */
    uint32_t index_count;            /* @20 */
    uint32_t first_index;            /* @24 */
    uint32_t first_instance;         /* @28 */
} tr_primitive_info;

/* `CullingPushConstants` (shared-structs/src/lib.rs:270-279; filled at src/main.rs:1728-1746). */
typedef struct tr_culling_push_constants {
    float view[16];                  /* @0  column-major view matrix */
    float frustum_x_xz[2];           /* @64 */
    float frustum_y_yz[2];           /* @72 */
    float z_near;                    /* @80 */
    float _pad[3];
} tr_culling_push_constants;

/* VkDrawIndexedIndirectCommand as demultiplex_draws writes it (shader/src/lib.rs:401-409; 20 B). */
typedef struct tr_draw_command {
    uint32_t index_count;
    uint32_t instance_count;
    uint32_t first_index;
    int32_t  vertex_offset;
    uint32_t first_instance;
} tr_draw_command;

#define TR_NUM_DRAW_BUFFERS 4u   /* shader/src/lib.rs:467 */

/* shared-structs/src/lib.rs:322 */
#define TR_MAX_LIGHTS_PER_CLUSTER 128u
#define TR_MAX_DEPTH_SLICES 64u   /* src/main.rs:62 uses 16 */

/* ----------------------------------------------- G-buffer ("TGB-v1" planes) */
/*
 * What the rasteriser hands the two fragment entry points (interpolated `position`,
 * `normal`, `uv`, flat `material_id`, flat `model_scale`, `frag_coord.z`;
 * shader/src/lib.rs:38-55, 165-181) laid out as structure-of-arrays planes in HBM,
 * row-major, plane index (y - origin_y)*width + (x - origin_x) for frame pixel (x, y).  The
 * frame size is push_constants.framebuffer_size; colour targets are always whole-frame buffers
 * (pitch = framebuffer_size.x).  All pointers are device pointers.
 */
typedef struct tr_gbuffer {
    const void* pos_depth;   /* float4: world position xyz, w = frag_coord.z (reversed-Z depth) */
    const void* nrm_scale;   /* float4: interpolated (un-normalised) normal xyz, w = model_scale */
    const void* uv;          /* float2: texture coordinates                                     */
    const void* material_id; /* uint32: index into materials; TR_NOT_COVERED = no fragment here */
    uint32_t    width;       /* plane size in pixels                                            */
    uint32_t    height;
    uint32_t    origin_x;    /* frame position of plane element (0,0): a rank that shades one    */
    uint32_t    origin_y;    /* screen tile holds only that tile's planes (0,0 = whole frame)    */
} tr_gbuffer;

#define TR_NOT_COVERED 0xFFFFFFFFu

/* shader/src/tonemapping.rs:29-39 `BakedLottesTonemapperParams` (push constants of fragment_tonemap). */
typedef struct tr_tonemap_params {
    float a, b, c, d;
    float crosstalk, saturation, cross_saturation;
} tr_tonemap_params;

/* The un-baked parameters (colstodian `LottesTonemapperParams`, un-vendored: its defaults are NOT in the
 * reference tree, so they are explicit inputs here; tr_lottes_defaults() gives the values of Lottes' talk). */
typedef struct tr_lottes_params {
    float contrast, shoulder, hdr_max, mid_in, mid_out;
    float crosstalk, saturation, cross_saturation;
} tr_lottes_params;

/* Half-open pixel rectangle [x0,x1) x [y0,y1) in frame coordinates: the screen tile one rank
 * shades.  Must lie inside both the frame and the G-buffer planes. */
typedef struct tr_rect {
    uint32_t x0, y0, x1, y1;
} tr_rect;

/* Colour-target storage format. The reference's targets are R16G16B16A16_SFLOAT
 * (src/render_passes.rs:27-41, src/main.rs:2370); RGBA32F exists for parity tests
 * that want to see the value before the half rounding. */
typedef enum tr_format {
    TR_FORMAT_RGBA16F = 0,
    TR_FORMAT_RGBA32F = 1,
    TR_FORMAT_RGBA8 = 2,    /* tr_allgather_frame only: the tonemapped 8-bit frame (tr_tonemap's output) */
    TR_FORMAT_RGB8 = 3      /* ... and without its constant alpha (tr_tonemap_rgb8's output): 3 bytes per pixel */
} tr_format;

/*
 * Opaque-colour pyramid (`opaque_sampled_hdr_framebuffer`, src/main.rs:383-402): one
 * device allocation of RGBA16F texels, levels tightly packed one after another
 * (level l is max(width>>l,1) x max(height>>l,1), row-major).  Level count follows
 * mip_levels_for_size (src/main.rs:2590-2592).
 */
#define TR_MAX_MIP_LEVELS 16u
typedef struct tr_pyramid {
    void*    texels;                             /* device pointer, RGBA16F */
    uint32_t width, height, levels;
    uint32_t level_offset[TR_MAX_MIP_LEVELS];    /* in texels, from `texels` */
} tr_pyramid;

/*
 * One image of the bindless `textures[]` array (set 0 binding 0): what `load_image_from_bytes`
 * (src/model_loading.rs:335-390) is given — decoded RGBA8 level 0 and whether the loader asked for
 * R8G8B8A8_SRGB (diffuse, emissive, specular colour; :233-291) or R8G8B8A8_UNORM.  The index of a texture is its
 * position in the array passed to tr_upload_textures; MaterialInfo.textures.* index it.
 */
typedef struct tr_texture_desc {
    const uint8_t* rgba8;      /* HOST pointer, width*height*4 bytes, row 0 first */
    uint32_t width, height;
    uint32_t srgb;             /* 1 = R8G8B8A8_SRGB, 0 = R8G8B8A8_UNORM */
    uint32_t _reserved;
} tr_texture_desc;

/* The packed mip chain of one material texture (levels tightly packed, RGBA8, level l is
 * max(width>>l,1) x max(height>>l,1)); level count = mip_levels_for_size (src/model_loading.rs:354). */
typedef struct tr_texture_layout {
    uint32_t width, height, levels, srgb;
    uint32_t level_offset[TR_MAX_MIP_LEVELS];    /* in texels */
    uint32_t total_texels;
} tr_texture_layout;

typedef struct tr_context tr_context; /* opaque */

/* ------------------------------------------------------------------ context */

/* Replaces Vulkan instance/device bring-up (src/main.rs:114-275) for this path. */
tr_status tr_context_create(int32_t device_ordinal, tr_context** out_ctx);
tr_status tr_context_destroy(tr_context* ctx);

const char* tr_status_string(tr_status status);
/* hipError_t of the last failing HIP call on this context (0 if none). */
int32_t     tr_last_hip_error(const tr_context* ctx);
uint32_t    tr_abi_version(void);

/* mip_levels_for_size (src/main.rs:2590-2592) and the packed layout above.
 * `texels` is left NULL; the size to allocate is returned through out_bytes.  It includes 8 bytes of
 * tail padding after the last texel: the sampler fetches texels in 16-byte pairs and may read (never
 * use) them.  The allocation must be at least out_bytes. */
tr_status tr_pyramid_layout(uint32_t width, uint32_t height, tr_pyramid* out_pyramid, size_t* out_bytes);

/* -------------------------------------------------------------------- tables */

/* materials[] storage buffer (set 0 binding 2; src/main.rs:715-760, filled by
 * src/model_loading.rs:231-333).  Host pointer; copied and pre-digested on `stream`.  Texture ids (!= -1) refer to
 * the array given to tr_upload_textures (either upload order; checked when a pass is launched). */
tr_status tr_upload_materials(tr_context* ctx, const tr_material_info* materials_host, uint32_t count, void* stream);

/* lights[] storage buffer (set 2 binding 0; src/main.rs:450-496). Host pointer. */
tr_status tr_upload_lights(tr_context* ctx, const tr_light* lights_host, uint32_t count, void* stream);
/* Rewrites lights [first, first + count) of the uploaded array in place — what the reference does per frame to its two
 * rotating spotlights through a mapped buffer (`light_buffers.lights.write_mapped`, src/main.rs:1244-1256).  Host pointer,
 * read before the call returns; the records travel inside kernel arguments (TR_UPDATE_MAX_BYTES per launch), so the call
 * allocates nothing, copies nothing from pageable memory and waits for nothing: it is ordered on `stream` like a launch —
 * and only there: a caller that shades on other streams of the same context orders them against this one itself (the full
 * uploads wait for launches of other streams; these do not).  The range must lie inside the last tr_upload_lights. */
#define TR_UPDATE_MAX_BYTES 3072u
tr_status tr_update_lights(tr_context* ctx, uint32_t first, uint32_t count, const tr_light* lights_host, void* stream);

/* cluster_light_counts / light_indices (set 2 bindings 1,2; src/main.rs:485-496):
 * device pointers, `num_clusters_total` u32 counts and num_clusters_total*128 u32
 * indices.  The context borrows them (e.g. the output of tr_assign_lights_to_clusters). */
tr_status tr_set_cluster_tables(tr_context* ctx, const void* cluster_light_counts_dev,
                                const void* light_indices_dev, uint32_t num_clusters_total);

/* ggx_lut.png as uploaded by src/main.rs:295-330 (R8G8B8A8_UNORM, row 0 first). Host pointer. */
tr_status tr_upload_ggx_lut(tr_context* ctx, const uint8_t* rgba8_host, uint32_t width, uint32_t height, void* stream);

/*
 * The bindless material textures (set 0 binding 0; src/model_loading.rs:160-215 pushes them in this order).
 * Replaces the whole array: level 0 of every image is copied to HBM and its full mip chain is generated on the
 * device on `stream` (the LINEAR vkCmdBlitImage chain of load_image_from_bytes; sRGB images are filtered in
 * linear light).  Sampling is the reference's `sampler` (src/main.rs:683-692): LINEAR min/mag/mip, REPEAT,
 * implicit LOD from the 2x2 quad's uv differences.  count == 0 clears the array.
 * Passes that shade a material with texture ids need: rect.x0 and rect.y0 even, rect.x1 / rect.y1 even or equal to
 * the frame size (whole quads), g->uv set, and every id < count (TR_ERR_INVALID_ARGUMENT otherwise).
 */
tr_status tr_upload_textures(tr_context* ctx, const tr_texture_desc* textures_host, uint32_t count, void* stream);

/* Layout of texture `index` as uploaded; tr_download_texture copies its whole chain (total_texels*4 bytes) back to
 * host memory after synchronising `stream` (inspection / tests). */
tr_status tr_texture_get_layout(const tr_context* ctx, uint32_t index, tr_texture_layout* out);
tr_status tr_download_texture(tr_context* ctx, uint32_t index, void* rgba8_host_out, size_t capacity_bytes, void* stream);

/* ------------------------------------------------------- frustum culling */

/*
 * `frustum_culling` (shader/src/lib.rs:411-465; dispatched at src/main.rs:1716-1763 after the count buffer is
 * zeroed, :1668-1674): instance_counts[p] = number of instances of primitive p whose transformed bounding sphere
 * is inside the view frustum.  All pointers are device pointers; instance_counts (num_primitives u32) is zeroed
 * by this call.
 */
tr_status tr_frustum_culling(tr_context* ctx, const void* primitives, uint32_t num_primitives, const void* instances,
                             uint32_t num_instances, const tr_culling_push_constants* push, void* instance_counts,
                             void* stream);

/*
 * `demultiplex_draws` (shader/src/lib.rs:469-517; src/main.rs:1811-1838): one tr_draw_command per primitive with
 * a non-zero instance count, appended to the draw buffer its draw_buffer_index names.  draw_counts: 4 u32
 * (zeroed by this call); draws[k]: device arrays with room for every primitive of that kind.  The reference
 * appends in atomic (arbitrary) order; here commands come out in ascending primitive order (deterministic).
 */
tr_status tr_demultiplex_draws(tr_context* ctx, const void* primitives, uint32_t num_primitives,
                               const void* instance_counts, void* draw_counts, void* const draws[TR_NUM_DRAW_BUFFERS],
                               void* stream);

/* ------------------------------------------------------- geometry front end */

/*
 * The model buffers of src/model_loading.rs / `ModelStagingBuffers` (vertex bindings 0-2 of src/pipelines.rs:287-304,
 * UINT32 indices, `primitives` and `instances` storage buffers).  HOST pointers; copied to HBM on `stream`, and the
 * per-frame work buffers of the rasteriser are sized from them.  Replaces any previous geometry.
 */
typedef struct tr_geometry_desc {
    const float* position;            /* 3 floats per vertex */
    const float* normal;              /* 3 floats per vertex */
    const float* uv;                  /* 2 floats per vertex */
    uint32_t num_vertices;
    const uint32_t* index;            /* triangle list */
    uint32_t num_indices;
    const tr_primitive_info* primitives;
    uint32_t num_primitives;
    const tr_instance* instances;
    uint32_t num_instances;
} tr_geometry_desc;
tr_status tr_upload_geometry(tr_context* ctx, const tr_geometry_desc* geometry_host, void* stream);
/* Rewrites instances [first, first + count) of the uploaded geometry in place — the reference's per-frame
 * `model_buffers.instances.write_mapped` of the rotating model (src/main.rs:1258-1261, 1316-1322).  Every record keeps its
 * primitive_id (TR_ERR_INVALID_ARGUMENT otherwise: the draw streams and the rasteriser's work buffers were sized from the
 * per-primitive instance ranges); nothing is re-allocated or re-sized.  Same transport and ordering as tr_update_lights. */
tr_status tr_update_instances(tr_context* ctx, uint32_t first, uint32_t count, const tr_instance* instances_host, void* stream);

/* One writable TGB-v1 layer covering the whole frame (push->framebuffer_size): device pointers.  Where a pixel has no
 * fragment only material_id (= TR_NOT_COVERED) is written; the other planes keep whatever they held. */
typedef struct tr_gbuffer_target {
    void* pos_depth;    /* float4 */
    void* nrm_scale;    /* float4 */
    void* uv;           /* float2 */
    void* material_id;  /* uint32 */
} tr_gbuffer_target;

/*
 * Replaces the depth pre-passes + EQUAL-tested colour-pass rasterisation of src/main.rs:1900-2042 (pipelines
 * src/pipelines.rs:309-398): runs the vertex stage (vertex_instanced_with_scale, shader/src/lib.rs:356-385) and
 * rasterises the four demultiplexed draw buffers of the uploaded geometry into the two layers the shading passes
 * read: `opaque` <- buffers 0 and 1 (nearest fragment, reversed-Z GREATER; buffer 1 with the alpha-clip kill of
 * depth_pre_pass_alpha_clip), `transmissive` <- buffers 2 and 3 where nearer than the opaque layer.
 * draw_counts / draws: device pointers as produced by tr_demultiplex_draws.  Materials (and textures, for alpha
 * clipping) must have been uploaded.  Fixed-function rules restated: see oracle/tr_oracle.h o_rasterize.
 * Stream capture: allowed — like tr_draw_scene and tr_record_frame, which rasterise through this — once a call with the same
 * frame size has run outside the capture (the scan's frame counter lives on the device and moves on with every replay).  What
 * cannot be captured is refused with TR_ERR_UNSUPPORTED before anything is enqueued: (re)allocating the visibility buffers
 * for another frame size, rebuilding a table, the timed frame recorder.  Replays of a captured frame must not interleave
 * with other rasterising calls of the context.  What a replay relies on, precisely: whether a call enqueues the clear of its
 * work buffers is decided from the context's state when the call is RECORDED.  A captured tr_record_frame / tr_draw_scene
 * always zeroes the instance counts it culls into (so a replay may follow a direct tr_frustum_culling on the context's own
 * counts); the visibility words (133 MB at 4K) are only zeroed when the context says they are dirty at capture time, so a
 * graph must be captured behind a complete frame and never replayed after a direct call that leaves them set — a
 * tr_rasterize / tr_draw_scene into RGBA16F frames whose shading passes were not run.  Re-capture after such a call.
 */
tr_status tr_rasterize(tr_context* ctx, const void* draw_counts, const void* const draws[TR_NUM_DRAW_BUFFERS],
                       const tr_push_constants* push, const tr_gbuffer_target* opaque,
                       const tr_gbuffer_target* transmissive, void* stream);

/* Culling + demultiplex + rasterisation of the uploaded geometry in one call (the per-frame sequence of
 * src/main.rs:1660-1838 followed by the draws), with the context's own count / draw buffers. */
tr_status tr_draw_scene(tr_context* ctx, const tr_culling_push_constants* culling, const tr_push_constants* push,
                        const tr_gbuffer_target* opaque, const tr_gbuffer_target* transmissive, void* stream);

/* ------------------------------------------------------- clustered-light build */

/*
 * `write_cluster_data` (shader/src/lib.rs:519-594; recorded at start-up and on resize, src/main.rs:832-840,
 * 1478-1517): the view-space AABB of every cluster of the num_clusters.x * num_clusters.y * num_depth_slices
 * grid in `uniforms`.  inverse_perspective is column-major (perspective_matrix.inverse()).
 * cluster_aabbs_out: device pointer to that many tr_cluster_aabb.
 */
tr_status tr_write_cluster_data(tr_context* ctx, const tr_uniforms* uniforms, const float inverse_perspective[16],
                                const uint32_t screen_dimensions[2], void* cluster_aabbs_out, void* stream);

/*
 * `assign_lights_to_clusters` (shader/src/lib.rs:596-645; every frame, src/main.rs:1765-1798) for the lights of
 * the last tr_upload_lights: view_matrix column-major, view_rotation = camera_rotation.inverse() as (x,y,z,w).
 * Writes counts[num_clusters] and indices[num_clusters * 128] (device pointers; pass them to
 * tr_set_cluster_tables).  Unlike the reference's atomics, every list comes out sorted by light index.
 */
tr_status tr_assign_lights_to_clusters(tr_context* ctx, const float view_matrix[16], const float view_rotation[4],
                                       const void* cluster_aabbs, uint32_t num_clusters, void* counts_out,
                                       void* indices_out, void* stream);

/* -------------------------------------------------------------------- passes */

/*
 * "main opaque" colour pass: `fragment` (shader/src/lib.rs:164-249) for every covered
 * pixel of `rect`; the same value goes to hdr_out and to level 0 of the pyramid
 * (lib.rs:247-248).  Pixels with material_id == TR_NOT_COVERED get the clear colour
 * (0,0,0,1) (src/main.rs:1592-1601).  hdr_out: width*height texels of `format`;
 * opaque_mip0_out is always RGBA16F (may be NULL to skip the second write).
 */
tr_status tr_shade_opaque(tr_context* ctx, const tr_gbuffer* gbuffer, const tr_uniforms* uniforms,
                          const tr_push_constants* push, void* hdr_out, tr_format format,
                          void* opaque_mip0_out, tr_rect rect, void* stream);
/* The same pass writing INTO THE PYRAMID: level 0 as above and — when the target is RGBA16F, both frame sizes are even and
 * the rect lies on even pixels — level 1 as well, from the 2x2 quads of the values the pass stores (a LINEAR blit of an
 * even-sized level is the box of its rounded texels: bit for bit what tr_generate_mips would write), so that the chain
 * never reads level 0 back (66 of its 88 MB at 4K).  *next_level_out = the level to continue from:
 * tr_generate_mips_from(ctx, pyramid, *next_level_out, stream) completes the pyramid (2, or 1 when level 1 was not written).
 * `pyramid` must have the frame's size (push->framebuffer_size). */
tr_status tr_shade_opaque_pyramid(tr_context* ctx, const tr_gbuffer* gbuffer, const tr_uniforms* uniforms,
                                  const tr_push_constants* push, void* hdr_out, tr_format format,
                                  const tr_pyramid* pyramid, tr_rect rect, uint32_t* next_level_out, void* stream);

/* "opaque framebuffer mipchain": generate_mips (src/main.rs:2046-2064): levels 1.. from level 0,
 * each level a LINEAR blit of the previous one, fp32 accumulate, RTNE store to RGBA16F. */
tr_status tr_generate_mips(tr_context* ctx, const tr_pyramid* pyramid, void* stream);
/* The same chain from level `first_level` on (>= 1): levels first_level .. from level first_level - 1, which the caller
 * has completed — e.g. 3 after the level-2 all-gather of a row-band sharded frame (below). */
tr_status tr_generate_mips_from(tr_context* ctx, const tr_pyramid* pyramid, uint32_t first_level, void* stream);
/* Row-band sharded full pipeline (no reference counterpart: src/main.rs:243 is one queue): levels 1 and 2
 * of the rows [y0, y1) of level 0 (a band on 4-row boundaries of a frame whose sizes are multiples of 4: the blits are
 * exact 2x2 boxes that never look outside the band).  TR_ERR_UNSUPPORTED otherwise (exchange all of level 0 instead). */
tr_status tr_generate_mips_band(tr_context* ctx, const tr_pyramid* pyramid, uint32_t y0, uint32_t y1, void* stream);
/* ... and what the transmissive pass of such a rank may sample: rows [row_lo, row_hi) of level 0 (its band + the halo it
 * received) and, halved, of level 1; levels >= 2 are whole on every rank.  With a window set tr_shade_transmission
 * records in *excess_word_dev (device, 4 bytes, zeroed by the caller) 1 + the largest number of level-0 rows by which a
 * tap of levels 0 / 1 lay outside it (atomicMax; 0 = the pass is exact) — the caller re-runs the exchange with a larger
 * halo (or all of level 0) and the pass when it is not 0.  row_hi = 0 or a NULL word: off. */
tr_status tr_set_tap_window(tr_context* ctx, uint32_t row_lo, uint32_t row_hi, void* excess_word_dev);

/*
 * "opaque transmissive objects": `fragment_transmission` (shader/src/lib.rs:37-162) for
 * every covered pixel of `rect`, overwriting hdr_inout (no blending, depth EQUAL:
 * src/pipelines.rs:338-347); uncovered pixels keep their value (attachment LOAD,
 * src/render_passes.rs:135-152).
 */
tr_status tr_shade_transmission(tr_context* ctx, const tr_gbuffer* gbuffer, const tr_uniforms* uniforms,
                                const tr_push_constants* push, const tr_pyramid* pyramid,
                                void* hdr_inout, tr_format format, tr_rect rect, void* stream);

/* ------------------------------------------------- the glam-pbr shading API */
/*
 * glam-pbr's public functions (glam-pbr/src/lib.rs), batched: element i of the output is the reference function
 * applied to element i of the input.  They are pure per-sample functions in the reference (by-value Copy structs,
 * no errors, NaN propagates); here the arrays live in HBM (device pointers, tightly packed, 4-byte aligned), one
 * thread evaluates one element on `stream`, and the arithmetic is the passes' own device code (same digest, same
 * light evaluation, same samplers), so a caller that keeps its own G-buffer gets the values the passes produce.
 * Direction vectors (`normal`, `view`, `light`) are unit vectors, as the reference's only caller passes them
 * (shader/src/lib.rs:79-80, shader/src/lighting.rs:229; `Normal` / `View` / `Light` wrap them without normalising).
 */

/* `MaterialParams` (:172-179; `PerceptualRoughness` and `IndexOfRefraction` are f32 newtypes). 40 B. */
typedef struct tr_material_params {
    float diffuse_colour[3];
    float metallic;
    float perceptual_roughness;
    float index_of_refraction;
    float specular_colour[3];
    float specular_factor;
} tr_material_params;

/* `BasicBrdfParams` (:163-169). 88 B. */
typedef struct tr_basic_brdf_params {
    float normal[3];
    float light[3];              /* direction to the light */
    float light_intensity[3];
    float view[3];
    tr_material_params material_params;
} tr_basic_brdf_params;

/* `BrdfResult` (:438-441). 24 B. */
typedef struct tr_brdf_result {
    float diffuse[3];
    float specular[3];
} tr_brdf_result;

/* The arguments of `transmission_btdf(material_params, normal, view, light)` (:200-205). 76 B. */
typedef struct tr_transmission_btdf_params {
    tr_material_params material_params;
    float normal[3];
    float view[3];
    float light[3];
} tr_transmission_btdf_params;

/* `IblVolumeRefractionParams` (:235-246). 168 B. */
typedef struct tr_ibl_volume_refraction_params {
    tr_material_params material_params;
    uint32_t framebuffer_size_x;
    float normal[3];
    float view[3];
    float proj_view_matrix[16];  /* column-major */
    float position[3];
    float thickness;
    float model_scale;
    float attenuation_distance;  /* +INF = no attenuation */
    float attenuation_colour[3];
} tr_ibl_volume_refraction_params;

/* What `light_direction_and_attenuation` returns (:12-23): (direction, distance, attenuation). 20 B. */
typedef struct tr_light_direction {
    float direction[3];
    float distance;
    float attenuation;
} tr_light_direction;

/* basic_brdf (:377-423): params_dev = tr_basic_brdf_params[count], results_dev = tr_brdf_result[count]. */
tr_status tr_basic_brdf(tr_context* ctx, const void* params_dev, uint32_t count, void* results_dev, void* stream);
/* transmission_btdf (:200-233): params_dev = tr_transmission_btdf_params[count], rgb_dev = float[count][3]. */
tr_status tr_transmission_btdf(tr_context* ctx, const void* params_dev, uint32_t count, void* rgb_dev, void* stream);
/* ibl_volume_refraction (:292-354) with the two closures the reference's caller passes (shader/src/lib.rs:126-138):
 * framebuffer_sampler = the opaque pyramid through clamp_sampler (trilinear, `framebuffer` below),
 * ggx_lut_sampler = the GGX LUT given to tr_upload_ggx_lut (bilinear, clamp).
 * params_dev = tr_ibl_volume_refraction_params[count], rgb_dev = float[count][3]. */
tr_status tr_ibl_volume_refraction(tr_context* ctx, const void* params_dev, uint32_t count, const tr_pyramid* framebuffer,
                                   void* rgb_dev, void* stream);
/* ibl_volume_refraction<FSamp, GSamp> with the CALLER's samplers (:292-299).  A closure cannot cross a C ABI onto the device,
 * so the generic function is offered as the two halves its closures cut it into:
 *   tr_ibl_volume_refraction_requests: what the function asks of its closures (:326-341) — requests_dev = float[count][5]:
 *       framebuffer_sampler's arguments texture_coords.x, texture_coords.y, framebuffer_lod, then ggx_lut_sampler's
 *       normal_dot_view, perceptual_roughness;
 *   tr_ibl_volume_refraction_resolve: the rest of the function (:338-353) given the closures' answers —
 *       framebuffer_rgb_dev = float[count][3] (Vec3 of framebuffer_sampler), lut_ab_dev = float[count][2] (Vec2 of
 *       ggx_lut_sampler) -> rgb_dev = float[count][3].
 * tr_ibl_volume_refraction above is this pair with the reference caller's own closures in between. */
tr_status tr_ibl_volume_refraction_requests(tr_context* ctx, const void* params_dev, uint32_t count, void* requests_dev,
                                            void* stream);
tr_status tr_ibl_volume_refraction_resolve(tr_context* ctx, const void* params_dev, uint32_t count,
                                           const void* framebuffer_rgb_dev, const void* lut_ab_dev, void* rgb_dev, void* stream);
/* light_direction_and_attenuation (:12-23): two float[count][3] arrays in, tr_light_direction[count] out. */
tr_status tr_light_direction_and_attenuation(tr_context* ctx, const void* fragment_position_dev,
                                             const void* light_position_dev, uint32_t count, void* out_dev, void* stream);
/* d_ggx (:101-109), v_smith_ggx_correlated (:114-133): float[count] arrays; `roughness` is the ACTUAL roughness. */
tr_status tr_d_ggx(tr_context* ctx, const void* normal_dot_halfway_dev, const void* roughness_dev, uint32_t count,
                   void* out_dev, void* stream);
tr_status tr_v_smith_ggx_correlated(tr_context* ctx, const void* normal_dot_view_dev, const void* normal_dot_light_dev,
                                    const void* roughness_dev, uint32_t count, void* out_dev, void* stream);
/* fresnel_schlick (:137-139): view_dot_halfway float[count], f0 / f90 / out float[count][3]. */
tr_status tr_fresnel_schlick(tr_context* ctx, const void* view_dot_halfway_dev, const void* f0_dev, const void* f90_dev,
                             uint32_t count, void* out_dev, void* stream);
/* compute_f0 (:454-465; exported by glam-pbr, unused by the renderer): metallic, ior float[count], diffuse_colour and
 * out float[count][3]. */
tr_status tr_compute_f0(tr_context* ctx, const void* metallic_dev, const void* index_of_refraction_dev,
                        const void* diffuse_colour_dev, uint32_t count, void* out_dev, void* stream);

/* `LightClusterCoefficients::get_depth_slice` (shared-structs/src/lib.rs:54-63; called per fragment at
 * shader/src/lib.rs:88-98, 205-215) over an array: frag_depth_dev = float[count] (`frag_coord.z`), slices_out_dev =
 * uint32_t[count].  Index work: BIT-EXACT with the reference's fp32 arithmetic (evaluated with the host's libm) for
 * every depth in [0, +inf] and NaN — the shading passes call the same device function for their cluster lookup.
 * Coefficients whose far-plane slice exceeds TR_MAX_DEPTH_SLICES return TR_ERR_UNSUPPORTED. */
tr_status tr_get_depth_slice(tr_context* ctx, const tr_light_cluster_coefficients* coefficients, const void* frag_depth_dev,
                             uint32_t count, void* slices_out_dev, void* stream);
/* Host only (no device needed): the table behind that exactness.  get_depth_slice is a non-increasing step function
 * of a non-negative depth; thresholds_out[k] (k = 1 .. *max_slice_out) receives the LARGEST depth whose slice is >= k,
 * found by bisection over bit patterns with the reference's own fp32 operations; thresholds_out[0] = +inf,
 * thresholds_out[*max_slice_out + 1] = -1.  So slice(d) = #{k >= 1 : d <= thresholds_out[k]} for d in [0, +inf].
 * thresholds_out must hold TR_MAX_DEPTH_SLICES + 2 floats. */
tr_status tr_depth_slice_thresholds(const tr_light_cluster_coefficients* coefficients, float* thresholds_out,
                                    uint32_t* max_slice_out);

/* ------------------------------------------------- multi-GPU: row bands + composite */
/*
 * New surface (the reference is one VkQueue, src/main.rs:243): SURVEY.md section 8e.  Pixels of both passes are
 * independent given replicated read-only inputs, so a frame is cut into contiguous ROW BANDS, one per rank (one
 * process per GPU).  No collective while shading; the exchanges are in-place all-gathers of whole bands over
 * RCCL / xGMI: the final composite (every rank ends with the whole frame) and, for the full opaque -> mips ->
 * transmissive pipeline, level 0 of the opaque pyramid between the two passes (the transmissive pass samples the
 * whole pyramid).
 *
 * RCCL is loaded at run time (dlopen "librccl.so.1") the first time one of these is called: the library has no
 * link-time dependency on it, and a process that already holds an RCCL (a host framework's) shares that copy.
 */
typedef struct tr_comm tr_comm;   /* opaque: one RCCL communicator of this rank */
#define TR_COMM_ID_BYTES 128u     /* = NCCL_UNIQUE_ID_BYTES */

/* Host only, no device needed: the band of `rank`.  Bands are `*rows_per_rank` = ceil(height / nranks) rounded up to
 * a multiple of 4 rows (the 16x4 wave tile; also keeps 2x2 quads whole), rows [*y0, *y1) clipped to the frame — the
 * last bands of a frame whose height does not divide may be short or empty (y0 == y1).  A composite buffer must
 * hold nranks * rows_per_rank rows (>= height): an all-gather moves equal counts. */
tr_status tr_band_rows(uint32_t height, uint32_t nranks, uint32_t rank, uint32_t* rows_per_rank, uint32_t* y0,
                       uint32_t* y1);
/* One rank (conventionally 0) creates the id; the application hands its 128 bytes to the other ranks over any host
 * channel; then EVERY rank calls tr_comm_create (collective, blocks until all nranks arrived) with its context's
 * device current. */
tr_status tr_comm_unique_id(uint8_t id_out[TR_COMM_ID_BYTES]);
tr_status tr_comm_create(tr_context* ctx, const uint8_t id[TR_COMM_ID_BYTES], uint32_t nranks, uint32_t rank,
                         tr_comm** out_comm);
/* Wraps a communicator the host already has (ncclComm_t, borrowed: tr_comm_destroy leaves it alone). */
tr_status tr_comm_from_nccl(void* nccl_comm, uint32_t nranks, uint32_t rank, tr_comm** out_comm);
tr_status tr_comm_destroy(tr_comm* comm);
/* ncclResult_t of the last failing RCCL call on this communicator (0 if none; -1: RCCL could not be loaded). */
int32_t   tr_comm_last_error(const tr_comm* comm);
/* What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): a host that wants to be sure its
 * composite runs over RCCL — and over how many ranks — asks here instead of trusting its own book-keeping. */
tr_status tr_comm_query(tr_comm* comm, uint32_t* nranks_out, uint32_t* rank_out);
/* The composite (and the mid-frame level-0 exchange): `frame_dev` holds nranks * rows_per_rank rows of `width`
 * pixels of `format`; this rank has written band `rank` (rows rank * rows_per_rank ...); on return (stream order)
 * every band is everywhere.  In place: ncclAllGather(sendbuff = recvbuff + rank * band_bytes).  Asynchronous on
 * `stream`; call it on every rank.  `format` may also be TR_FORMAT_RGBA8: the composite of the frame as it is presented
 * (each rank tonemaps its band first) moves half the bytes of the RGBA16F one. */
tr_status tr_allgather_frame(tr_context* ctx, tr_comm* comm, void* frame_dev, uint32_t width, uint32_t rows_per_rank,
                             tr_format format, void* stream);

/* The mid-frame exchange of a row-band sharded full pipeline WITHOUT gathering all of level 0: `level_rows_dev` is a
 * whole pyramid level (`total_rows` rows of `row_bytes` bytes) of which this rank has written its band (rows
 * rank * rows_per_rank ..., clipped); on return (stream order) it also holds the `halo_rows` rows above and below its
 * band, received from the ranks that own them (ncclSend / ncclRecv, one group; halo_rows >= total_rows: an all-gather).
 * Per link a rank's band-border rows instead of its whole band.  Call it on every rank with the same arguments.
 * tr_halo_rows (host only): the rows [*y0, *y1) of `owner`'s band that `reader` receives (y0 == y1: none). */
tr_status tr_exchange_halo(tr_context* ctx, tr_comm* comm, void* level_rows_dev, uint32_t row_bytes, uint32_t total_rows,
                           uint32_t rows_per_rank, uint32_t halo_rows, void* stream);
tr_status tr_halo_rows(uint32_t total_rows, uint32_t rows_per_rank, uint32_t nranks, uint32_t owner, uint32_t reader,
                       uint32_t halo_rows, uint32_t* y0, uint32_t* y1);

/* Rank-interleaved strips, for frames whose cost is uneven over the screen (sky above, geometry below: contiguous bands
 * leave the ranks of the sky idle): the frame is cut into strips of `strip_rows` rows (a multiple of 4; 64 is a good
 * size), strip s belongs to rank s % nranks.  After tr_set_strips, tr_shade_opaque / tr_shade_transmission called with a
 * rect that spans the whole frame height (and G-buffer planes that cover it) shade THIS RANK'S STRIPS in one launch
 * each, in place: colour targets are whole-frame buffers and every strip lands at its own rows.  tr_set_strips(ctx, 0,
 * 1, 0) turns it off.  Any material set (one launch shades every material class); not combined with tr_record_frame
 * (TR_ERR_UNSUPPORTED).  SURVEY.md 8e; the reference is single-GPU (src/main.rs:243). */
tr_status tr_set_strips(tr_context* ctx, uint32_t strip_rows, uint32_t nranks, uint32_t rank);
/* Host only: rows [*y0, *y1) of the k-th strip of `rank` (strip k * nranks + rank of the frame), clipped to the frame;
 * y0 == y1 once the rank has no k-th strip. */
tr_status tr_strip_of_rank(uint32_t height, uint32_t strip_rows, uint32_t nranks, uint32_t rank, uint32_t k, uint32_t* y0,
                           uint32_t* y1);
/* The composite (and the level-0 exchange) of a strip-sharded frame: `frame_dev` is the whole frame (`height` rows of
 * `width` pixels of `format`), every rank has written its own strips in place; on return (stream order) every strip is
 * everywhere.  One RCCL group of in-place broadcasts, one per strip, rooted at the strip's rank.  Call it on every rank. */
tr_status tr_allgather_strips(tr_context* ctx, tr_comm* comm, void* frame_dev, uint32_t width, uint32_t height,
                              uint32_t strip_rows, tr_format format, void* stream);

/* ------------------------------------------------------------------ tonemap */

/* Host only: Lottes' curve constants from the un-baked parameters (what colstodian's
 * `BakedLottesTonemapperParams::from` computes for src/main.rs:506-510). */
tr_status tr_lottes_defaults(tr_lottes_params* out);
tr_status tr_bake_lottes_params(const tr_lottes_params* params, tr_tonemap_params* out);

/* "tonemapping": fragment_tonemap (shader/src/lib.rs:683-697, shader/src/tonemapping.rs) over the whole
 * RGBA16F frame, written as 8-bit sRGB like the reference's B8G8R8A8_SRGB swapchain (src/main.rs:175);
 * `bgra` != 0 selects that byte order, 0 gives R,G,B,A. */
tr_status tr_tonemap(tr_context* ctx, const void* hdr_rgba16f, uint32_t width, uint32_t height,
                     const tr_tonemap_params* params, void* out_rgba8, int32_t bgra, void* stream);
/* The same pixels without the alpha byte (it is 255 everywhere: fragment_tonemap writes alpha 1): 3 bytes per pixel,
 * r g b (or b g r), rows tightly packed — what a row-band sharded frame composites (tr_allgather_frame with
 * TR_FORMAT_RGB8): a quarter less per xGMI link than RGBA8.  width * height must be a multiple of 4. */
tr_status tr_tonemap_rgb8(tr_context* ctx, const void* hdr_rgba16f, uint32_t width, uint32_t height,
                          const tr_tonemap_params* params, void* out_rgb8, int32_t bgra, void* stream);

/* ------------------------------------------------------------------ one frame */

/*
 * Everything `record()` (src/main.rs:1551-2263) enqueues per frame for this path, in its order, as ONE call on one
 * stream: zero counters + frustum culling (:1660-1763) -> light assignment (:1765-1798) -> draw demultiplex
 * (:1811-1838) -> depth pre-passes + rasterisation into the two layers (:1900-2042) -> main opaque colour pass
 * (:1969-2001) -> opaque mip chain (:2046-2064) -> transmissive pass (:2066-2124) -> tonemap (:2126-2180).
 * Uses the context's uploaded geometry, materials, textures, lights and GGX LUT.  Every pointer in the descriptor
 * is a device pointer owned by the caller unless marked host; nothing is allocated per frame.
 */
typedef struct tr_frame_desc {
    const tr_push_constants* push;                 /* host */
    const tr_uniforms* uniforms;                   /* host */
    const tr_culling_push_constants* culling;      /* host */
    const float* view_matrix;                      /* host, 16 floats column-major (AssignLightsPushConstants) */
    const float* view_rotation;                    /* host, quaternion xyzw = camera_rotation.inverse() */
    const void* cluster_aabbs;                     /* from tr_write_cluster_data (start-up / resize) */
    uint32_t num_clusters;
    uint32_t _reserved;
    void* cluster_light_counts;                    /* out: num_clusters u32; bound as the passes' tables */
    void* light_indices;                           /* out: num_clusters * 128 u32 */
    tr_gbuffer_target opaque_layer;                /* work: whole-frame TGB-v1 planes.  Scratch: an RGBA16F frame is shaded
                                                      straight from the rasteriser's visibility words and leaves them
                                                      untouched; an RGBA32F frame resolves into them */
    tr_gbuffer_target transmissive_layer;
    tr_pyramid pyramid;                            /* work: opaque_sampled_hdr_framebuffer */
    void* hdr;                                     /* out: the HDR colour target, whole frame */
    tr_format hdr_format;
    int32_t bgra;                                  /* byte order of ldr_out (see tr_tonemap) */
    const tr_tonemap_params* tonemap;              /* host; may be NULL together with ldr_out: no tonemap */
    void* ldr_out;                                 /* out: width*height RGBA8 (sRGB encoded), or NULL */
} tr_frame_desc;
tr_status tr_record_frame(tr_context* ctx, const tr_frame_desc* frame, void* stream);

/* The same frame with a GPU timestamp pair around every pass: the reference's Tracy GPU zones (Vulkan timestamp queries,
 * src/profiling.rs:134-236) under the names it gives them in `record()` (src/main.rs:1643, 1653, 1831, 1902, 1989,
 * 2048, 2094, 2227): "all commands", "frustum culling", "demultiplex draws compute shader", "depth pre pass" (here:
 * the visibility-buffer rasteriser that stands for the three depth pre-passes and the EQUAL-tested draws), "main opaque",
 * "opaque framebuffer mipchain", "opaque transmissive objects", "tonemapping"; plus "assign lights to clusters", which
 * the reference records without a zone (:1765-1798).  A measuring call: it records HIP events on `stream`, WAITS for
 * the frame, and writes up to `capacity` zones (host memory; `name` points to a static string) and their count. */
typedef struct tr_frame_zone {
    const char* name;
    float milliseconds;
    uint32_t _pad;
} tr_frame_zone;
#define TR_MAX_FRAME_ZONES 16u
tr_status tr_record_frame_timed(tr_context* ctx, const tr_frame_desc* frame, void* stream, tr_frame_zone* zones_out,
                                uint32_t capacity, uint32_t* num_zones_out);

#ifdef __cplusplus
} /* extern "C" */
#endif

#ifdef __cplusplus
#define TR_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define TR_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
TR_STATIC_ASSERT(sizeof(tr_material_params) == 40, "MaterialParams is 10 floats");
TR_STATIC_ASSERT(sizeof(tr_basic_brdf_params) == 88, "BasicBrdfParams is 22 floats");
TR_STATIC_ASSERT(sizeof(tr_brdf_result) == 24, "BrdfResult is 6 floats");
TR_STATIC_ASSERT(sizeof(tr_transmission_btdf_params) == 76, "transmission_btdf arguments are 19 floats");
TR_STATIC_ASSERT(sizeof(tr_ibl_volume_refraction_params) == 168, "IblVolumeRefractionParams is 42 words");
TR_STATIC_ASSERT(sizeof(tr_light_direction) == 20, "(Vec3, f32, f32)");
TR_STATIC_ASSERT(sizeof(tr_push_constants) == 96, "PushConstants is 96 B");
TR_STATIC_ASSERT(offsetof(tr_push_constants, view_position) == 64, "view_position @64");
TR_STATIC_ASSERT(offsetof(tr_push_constants, framebuffer_size) == 80, "framebuffer_size @80");
TR_STATIC_ASSERT(offsetof(tr_push_constants, acceleration_structure_address) == 88, "as address @88");
TR_STATIC_ASSERT(sizeof(tr_uniforms) == 96, "Uniforms is 96 B");
TR_STATIC_ASSERT(offsetof(tr_uniforms, sun_dir) == 32, "sun_dir @32");
TR_STATIC_ASSERT(offsetof(tr_uniforms, sun_intensity) == 48, "sun_intensity @48");
TR_STATIC_ASSERT(offsetof(tr_uniforms, cluster_size_in_pixels) == 64, "cluster_size_in_pixels @64");
TR_STATIC_ASSERT(offsetof(tr_uniforms, num_clusters) == 72, "num_clusters @72");
TR_STATIC_ASSERT(offsetof(tr_uniforms, debug_clusters) == 80, "debug_clusters @80");
TR_STATIC_ASSERT(offsetof(tr_uniforms, ggx_lut_texture_index) == 84, "ggx_lut_texture_index @84");
TR_STATIC_ASSERT(sizeof(tr_material_info) == 160, "MaterialInfo stride is 160 B");
TR_STATIC_ASSERT(offsetof(tr_material_info, metallic_factor) == 36, "metallic_factor @36");
TR_STATIC_ASSERT(offsetof(tr_material_info, diffuse_factor) == 48, "diffuse_factor @48");
TR_STATIC_ASSERT(offsetof(tr_material_info, emissive_factor) == 64, "emissive_factor @64");
TR_STATIC_ASSERT(offsetof(tr_material_info, normal_map_scale) == 80, "normal_map_scale @80");
TR_STATIC_ASSERT(offsetof(tr_material_info, index_of_refraction) == 88, "index_of_refraction @88");
TR_STATIC_ASSERT(offsetof(tr_material_info, attenuation_distance) == 100, "attenuation_distance @100");
TR_STATIC_ASSERT(offsetof(tr_material_info, attenuation_colour) == 112, "attenuation_colour @112");
TR_STATIC_ASSERT(offsetof(tr_material_info, specular_factor) == 128, "specular_factor @128");
TR_STATIC_ASSERT(offsetof(tr_material_info, specular_colour_factor) == 144, "specular_colour_factor @144");
TR_STATIC_ASSERT(sizeof(tr_instance) == 48 && offsetof(tr_instance, primitive_id) == 32, "Instance stride is 48 B");
TR_STATIC_ASSERT(sizeof(tr_primitive_info) == 32 && offsetof(tr_primitive_info, draw_buffer_index) == 16, "PrimitiveInfo is 32 B");
TR_STATIC_ASSERT(offsetof(tr_culling_push_constants, frustum_x_xz) == 64 && offsetof(tr_culling_push_constants, z_near) == 80, "CullingPushConstants");
TR_STATIC_ASSERT(sizeof(tr_draw_command) == 20, "VkDrawIndexedIndirectCommand is 20 B");
TR_STATIC_ASSERT(sizeof(tr_light) == 48, "Light stride is 48 B");
TR_STATIC_ASSERT(sizeof(tr_cluster_aabb) == 32, "ClusterAabb is 32 B");

#endif /* TR_SHADE_H */
