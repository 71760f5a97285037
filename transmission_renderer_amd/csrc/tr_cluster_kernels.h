// tr_cluster_kernels.h — the clustered-light build that feeds the shading kernels.
//
//   write_cluster_data          shader/src/lib.rs:519-594   one view-space AABB per cluster (init / resize only)
//   assign_lights_to_clusters   shader/src/lib.rs:596-645   per frame: which lights touch which cluster
//
// The reference launches one invocation per (cluster, light) pair and appends with an atomic counter, so its
// per-cluster lists come out in arbitrary order.  Here ONE WAVE owns a cluster: its 64 lanes test 64 lights at a
// time, a ballot gives the survivors and a prefix pop-count (mbcnt) their slots — the list is written sorted by
// light index with no atomics, deterministic from run to run (the shading kernels sum lights in list order).
// All arithmetic is IEEE fp32 with contraction off, in the reference's order: the outputs are bit-identical to
// the CPU oracle, which is bit-identical to the reference's compiled .spv (tests/golden/spirv_clusters.npz).
// Transcendentals (powf for the slice depths, cos/sin of the spot angle) are evaluated on the host with libm.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tr_shade.h"

namespace tr {

struct tr_cluster_build_params {
    float inverse_perspective[16];   // column-major
    float cluster_size_px[2];
    float screen_dims[2];            // screen_dimensions.as_vec2()
    uint32_t nx, ny, nz;
    float slice_depth[TR_MAX_DEPTH_SLICES + 1];  // slice_to_depth(0..nz), shared-structs/src/lib.rs:65-67
};

// Light as assign_lights_to_clusters reads it (one 48-byte record, like the reference's).
struct alignas(16) tr_alight {
    float pos[3];      float falloff_distance_sq;
    float spot_dir[3]; uint32_t is_spot;
    float cos_angle, sin_angle, _pad[2];   // of spotlight_direction_and_outer_angle.w
};
static_assert(sizeof(tr_alight) == 48, "48 B");

struct tr_assign_params {
    float view_matrix[16];   // column-major
    float view_rotation[4];  // quaternion x, y, z, w (camera_rotation.inverse(), src/main.rs:1788)
    uint32_t num_lights, num_clusters;
};

__device__ __forceinline__ void mat4_mul_vec4(const float* m, float x, float y, float z, float w, float out[4]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // glam 0.19: ((X*x + Y*y) + Z*z) + W*w
        float acc = m[0 + r] * x;
        acc = m[4 + r] * y + acc;
        acc = m[8 + r] * z + acc;
        acc = m[12 + r] * w + acc;
        out[r] = acc;
    }
}

// shader/src/lib.rs:519-580: one thread per cluster.
__global__ __launch_bounds__(64) void write_cluster_data_kernel(const tr_cluster_build_params p,
                                                                tr_cluster_aabb* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t id = blockIdx.x * 64u + threadIdx.x;
    const uint32_t total = p.nx * p.ny * p.nz;
    if (id >= total) return;
    const uint32_t x = id % p.nx, y = (id / p.nx) % p.ny, z = id / (p.nx * p.ny);
    float vs[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float sx = (float)(x + (uint32_t)k) * p.cluster_size_px[0], sy = (float)(y + (uint32_t)k) * p.cluster_size_px[1];
        float cx = sx / p.screen_dims[0], cy = sy / p.screen_dims[1];   // screen_to_clip :540-544
        cx = cx * 2.0f - 1.0f;
        cy = cy * 2.0f - 1.0f;
        float v[4];
        mat4_mul_vec4(p.inverse_perspective, cx, cy, 0.0f, 1.0f, v);     // clip_to_view :546-550
        vs[k][0] = v[0] / v[3];
        vs[k][1] = v[1] / v[3];
        vs[k][2] = v[2] / v[3];
    }
    const float zs[2] = {p.slice_depth[z], p.slice_depth[z + 1u]};
    float mn[3], mx[3];
    bool first = true;
#pragma unroll
    for (int k = 0; k < 2; ++k) {        // min point, then max point
#pragma unroll
        for (int s = 0; s < 2; ++s) {    // near, then far
            // line_intersection_to_z_plane(eye = (0,0,1), b, z) :582-594
            float ax = 0.0f, ay = 0.0f, az = 1.0f;
            float dx = vs[k][0] - ax, dy = vs[k][1] - ay, dz = vs[k][2] - az;
            float na = (0.0f * ax + 0.0f * ay) + 1.0f * az;
            float nd = (0.0f * dx + 0.0f * dy) + 1.0f * dz;
            float t = (zs[s] - na) / nd;
            float pt[3] = {ax + t * dx, ay + t * dy, az + t * dz};
            // reference order of the min/max chain: min_near, min_far, max_near, max_far
            (void)first;
            if (k == 0 && s == 0) {
                for (int c = 0; c < 3; ++c) mn[c] = mx[c] = pt[c];
            } else {
                for (int c = 0; c < 3; ++c) {
                    mn[c] = fminf(mn[c], pt[c]);
                    mx[c] = fmaxf(mx[c], pt[c]);
                }
            }
        }
    }
    tr_cluster_aabb o;
    o.min[0] = mn[0]; o.min[1] = mn[1]; o.min[2] = mn[2]; o._pad0 = 0.0f;
    o.max[0] = mx[0]; o.max[1] = mx[1]; o.max[2] = mx[2]; o._pad1 = 0.0f;
    out[id] = o;
}

// shader/src/lib.rs:596-645 + ClusterAabb::{distance_sq, cull_spotlight} shared-structs/src/lib.rs:290-319.
// One wave per cluster; block = 4 waves = 4 clusters.
__device__ __forceinline__ void assign_lights_body(const tr_assign_params& p, const tr_alight* __restrict__ lights,
                                                   const tr_cluster_aabb* __restrict__ clusters, uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ indices, uint32_t block) {
#pragma clang fp contract(off)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cluster = block * 4u + (threadIdx.x >> 6);
    if (cluster >= p.num_clusters) return;
    const tr_cluster_aabb box = clusters[cluster];
    // per-cluster constants of cull_spotlight
    const float cxm = (box.min[0] + box.max[0]) / 2.0f, cym = (box.min[1] + box.max[1]) / 2.0f,
                czm = (box.min[2] + box.max[2]) / 2.0f;
    const float rx = box.max[0] - cxm, ry = box.max[1] - cym, rz = box.max[2] - czm;
    const float radius = __fsqrt_rn((rx * rx + ry * ry) + rz * rz);
    uint32_t count = 0;
    uint32_t* list = indices + (size_t)cluster * TR_MAX_LIGHTS_PER_CLUSTER;
    for (uint32_t base = 0; base < p.num_lights; base += 64u) {
        const uint32_t li = base + lane;
        bool keep = false;
        if (li < p.num_lights) {
            const tr_alight L = lights[li];
            float lp[4];
            mat4_mul_vec4(p.view_matrix, L.pos[0], L.pos[1], L.pos[2], 1.0f, lp);
            // distance_sq: ((min - p).max(p - max)).max(0).length_squared()
            float dx = fmaxf(fmaxf(box.min[0] - lp[0], lp[0] - box.max[0]), 0.0f);
            float dy = fmaxf(fmaxf(box.min[1] - lp[1], lp[1] - box.max[1]), 0.0f);
            float dz = fmaxf(fmaxf(box.min[2] - lp[2], lp[2] - box.max[2]), 0.0f);
            float d2 = (dx * dx + dy * dy) + dz * dz;
            keep = !(d2 > L.falloff_distance_sq);
            if (keep && L.is_spot) {
                // view_rotation * spot_dir (glam 0.19 scalar Quat * Vec3)
                const float qx = p.view_rotation[0], qy = p.view_rotation[1], qz = p.view_rotation[2], qw = p.view_rotation[3];
                const float vx = L.spot_dir[0], vy = L.spot_dir[1], vz = L.spot_dir[2];
                const float b2 = (qx * qx + qy * qy) + qz * qz;
                const float s0 = qw * qw - b2;
                const float s1 = ((vx * qx + vy * qy) + vz * qz) * 2.0f;
                const float s2 = qw * 2.0f;
                const float crx = qy * vz - vy * qz, cry = qz * vx - vz * qx, crz = qx * vy - vx * qy;   // b.cross(v)
                const float dirx = (vx * s0 + qx * s1) + crx * s2, diry = (vy * s0 + qy * s1) + cry * s2,
                            dirz = (vz * s0 + qz * s1) + crz * s2;
                // cull_spotlight(origin = light position, direction, angle, range = falloff_distance_sq (sic))
                const float wx = cxm - lp[0], wy = cym - lp[1], wz = czm - lp[2];
                const float len_sq = (wx * wx + wy * wy) + wz * wz;
                const float v1 = (wx * dirx + wy * diry) + wz * dirz;
                const float v1sq = v1 * v1;
                const float closest = L.cos_angle * __fsqrt_rn(len_sq - v1sq) - v1 * L.sin_angle;
                const bool cull = (closest > radius) || (v1 > radius + L.falloff_distance_sq) || (v1 < -radius);
                keep = !cull;
            }
        }
        const uint64_t mask = __ballot(keep);
        if (keep) {
            const uint32_t slot = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (slot < TR_MAX_LIGHTS_PER_CLUSTER) list[slot] = li;
        }
        count += (uint32_t)__popcll(mask);
    }
    if (lane == 0) counts[cluster] = count < TR_MAX_LIGHTS_PER_CLUSTER ? count : TR_MAX_LIGHTS_PER_CLUSTER;
}
__global__ __launch_bounds__(256) void assign_lights_kernel(const tr_assign_params p, const tr_alight* __restrict__ lights,
                                                            const tr_cluster_aabb* __restrict__ clusters,
                                                            uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ indices) {
    assign_lights_body(p, lights, clusters, counts, indices, blockIdx.x);
}

}  // namespace tr
