/* host_frame.c — a frame through the C ABI (include/tr_shade.h) from a host that is plain C11: no Python, no torch, no C++.
 *
 * What the reference's Rust host would do after dropping ash / Vulkan, restated in C because
 * the image has no cargo: context -> uploads (materials, textures, lights, GGX LUT, geometry) -> tr_write_cluster_data
 * (start-up) -> per frame: tr_update_instances + tr_update_lights (the reference's mapped-buffer writes,
 * src/main.rs:1244-1261, 1316-1322) and ONE tr_record_frame (its record(), src/main.rs:1551-2263) -> the presented RGBA8
 * frame written to a file.  Device memory comes straight from the HIP runtime (hipMalloc); nothing else is linked.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/host_frame.c \
 *       -Ltransmission_renderer_amd -ltr_shade -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../transmission_renderer_amd' -o host_frame
 *   ./host_frame scene.bin frame.rgba8 [frames]
 *
 * scene.bin (little endian, written by tests/test_gpu_c_host.py from the same arrays the Python path uploads):
 *   u32 magic 'TRSC', u32 width, u32 height, u32 counts[9] = materials, lights, vertices, indices, primitives, instances,
 *   textures, lut_w, lut_h; then tr_push_constants, tr_uniforms, tr_culling_push_constants, float view[16], float
 *   view_rotation[4], float inverse_perspective[16], tr_lottes_params; then the arrays in the order of the counts
 *   (materials, lights, positions, normals, uvs, indices, primitives, instances), per texture {u32 w, h, srgb, pad; texels},
 *   the LUT's RGBA8 texels; then per frame k >= 1: u32 first_instance, u32 n_instances, instances, u32 first_light,
 *   u32 n_lights, lights — the records frame k rewrites before it is recorded.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tr_shade.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define TR_OK_(x) do { tr_status s_ = (x); if (s_ != TR_OK) { fprintf(stderr, "%s: %s (hip error %d)\n", #x, tr_status_string(s_), tr_last_hip_error(ctx)); exit(3); } } while (0)

static void* slurp(const char* path, size_t* size) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    *size = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(*size);
    if (fread(p, 1, *size, f) != *size) { perror("read"); exit(1); }
    fclose(f);
    return p;
}

static const uint8_t* cur;
static const void* take(size_t bytes) {
    const void* p = cur;
    cur += (bytes + 3u) & ~(size_t)3u;
    return p;
}

static void* dev(size_t bytes) {
    void* p = NULL;
    HIP_OK(hipMalloc(&p, bytes ? bytes : 4));
    HIP_OK(hipMemset(p, 0, bytes ? bytes : 4));
    return p;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s scene.bin frame.rgba8 [frames]\n", argv[0]); return 1; }
    const int frames = argc > 3 ? atoi(argv[3]) : 3;
    size_t size;
    const uint8_t* file = slurp(argv[1], &size);
    cur = file;
    const uint32_t* head = take(12 * 4);
    if (head[0] != 0x43535254u) { fprintf(stderr, "not a scene file\n"); return 1; }
    const uint32_t w = head[1], h = head[2], n_mat = head[3], n_light = head[4], n_vert = head[5], n_idx = head[6], n_prim = head[7],
                   n_inst = head[8], n_tex = head[9], lut_w = head[10], lut_h = head[11];
    const tr_push_constants* push = take(sizeof *push);
    const tr_uniforms* uniforms = take(sizeof *uniforms);
    const tr_culling_push_constants* culling = take(sizeof *culling);
    const float* view = take(64);
    const float* view_rotation = take(16);
    const float* inverse_perspective = take(64);
    const tr_lottes_params* lottes = take(sizeof *lottes);
    const tr_material_info* materials = take(sizeof(tr_material_info) * n_mat);
    const tr_light* lights = take(sizeof(tr_light) * n_light);
    tr_geometry_desc geo;
    memset(&geo, 0, sizeof geo);
    geo.position = take(12u * n_vert);
    geo.normal = take(12u * n_vert);
    geo.uv = take(8u * n_vert);
    geo.num_vertices = n_vert;
    geo.index = take(4u * n_idx);
    geo.num_indices = n_idx;
    geo.primitives = take(sizeof(tr_primitive_info) * n_prim);
    geo.num_primitives = n_prim;
    geo.instances = take(sizeof(tr_instance) * n_inst);
    geo.num_instances = n_inst;
    tr_texture_desc* textures = calloc(n_tex ? n_tex : 1, sizeof *textures);
    for (uint32_t i = 0; i < n_tex; ++i) {
        const uint32_t* t = take(16);
        textures[i].width = t[0];
        textures[i].height = t[1];
        textures[i].srgb = t[2];
        textures[i].rgba8 = take((size_t)t[0] * t[1] * 4u);
    }
    const uint8_t* lut = take((size_t)lut_w * lut_h * 4u);

    tr_context* ctx = NULL;
    {
        tr_status s = tr_context_create(0, &ctx);
        if (s != TR_OK) { fprintf(stderr, "tr_context_create: %s\n", tr_status_string(s)); return 3; }
    }
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    TR_OK_(tr_upload_materials(ctx, materials, n_mat, stream));
    TR_OK_(tr_upload_textures(ctx, n_tex ? textures : NULL, n_tex, stream));
    TR_OK_(tr_upload_lights(ctx, lights, n_light, stream));
    TR_OK_(tr_upload_ggx_lut(ctx, lut, lut_w, lut_h, stream));
    TR_OK_(tr_upload_geometry(ctx, &geo, stream));

    /* start-up (src/main.rs:832-840): the cluster AABBs of the grid the uniforms describe */
    const uint32_t num_clusters = uniforms->num_clusters[0] * uniforms->num_clusters[1] * uniforms->light_clustering_coefficients.num_depth_slices;
    void* aabbs = dev((size_t)num_clusters * sizeof(tr_cluster_aabb));
    const uint32_t screen[2] = {w, h};
    TR_OK_(tr_write_cluster_data(ctx, uniforms, inverse_perspective, screen, aabbs, stream));

    /* the frame's work buffers, allocated once */
    const size_t px = (size_t)w * h;
    tr_frame_desc fd;
    memset(&fd, 0, sizeof fd);
    fd.push = push;
    fd.uniforms = uniforms;
    fd.culling = culling;
    fd.view_matrix = view;
    fd.view_rotation = view_rotation;
    fd.cluster_aabbs = aabbs;
    fd.num_clusters = num_clusters;
    fd.cluster_light_counts = dev((size_t)num_clusters * 4u);
    fd.light_indices = dev((size_t)num_clusters * TR_MAX_LIGHTS_PER_CLUSTER * 4u);
    tr_gbuffer_target* layers[2] = {&fd.opaque_layer, &fd.transmissive_layer};
    for (int k = 0; k < 2; ++k) {
        layers[k]->pos_depth = dev(px * 16u);
        layers[k]->nrm_scale = dev(px * 16u);
        layers[k]->uv = dev(px * 8u);
        layers[k]->material_id = dev(px * 4u);
    }
    size_t pyramid_bytes = 0;
    TR_OK_(tr_pyramid_layout(w, h, &fd.pyramid, &pyramid_bytes));
    fd.pyramid.texels = dev(pyramid_bytes);
    fd.hdr = dev(px * 8u);
    fd.hdr_format = TR_FORMAT_RGBA16F;
    fd.bgra = 0;
    tr_tonemap_params tonemap;
    TR_OK_(tr_bake_lottes_params(lottes, &tonemap));
    fd.tonemap = &tonemap;
    fd.ldr_out = dev(px * 4u);

    for (int k = 0; k < frames; ++k) {
        if (k >= 1 && (size_t)(cur - file) + 8u <= size) {   /* this frame's rewrites, if the file holds any */
            const uint32_t* ir = take(8);
            const tr_instance* inst = take(sizeof(tr_instance) * ir[1]);
            const uint32_t* lr = take(8);
            const tr_light* lt = take(sizeof(tr_light) * lr[1]);
            TR_OK_(tr_update_instances(ctx, ir[0], ir[1], inst, stream));
            TR_OK_(tr_update_lights(ctx, lr[0], lr[1], lt, stream));
        }
        TR_OK_(tr_record_frame(ctx, &fd, stream));
    }
    uint8_t* out = malloc(px * 4u);
    HIP_OK(hipMemcpyAsync(out, fd.ldr_out, px * 4u, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    FILE* f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 1, px * 4u, f) != px * 4u) { perror(argv[2]); return 1; }
    fclose(f);
    /* and the HDR target beside it, for a bit-for-bit comparison of the un-tonemapped frame */
    char hdr_path[4096];
    snprintf(hdr_path, sizeof hdr_path, "%s.hdr16", argv[2]);
    uint8_t* hdr = malloc(px * 8u);
    HIP_OK(hipMemcpy(hdr, fd.hdr, px * 8u, hipMemcpyDeviceToHost));
    f = fopen(hdr_path, "wb");
    if (!f || fwrite(hdr, 1, px * 8u, f) != px * 8u) { perror(hdr_path); return 1; }
    fclose(f);
    printf("host_frame: %d frame(s) of %ux%u, ABI version %u\n", frames, w, h, tr_abi_version());
    TR_OK_(tr_context_destroy(ctx));
    return 0;
}
