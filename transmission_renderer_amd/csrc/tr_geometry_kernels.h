// tr_geometry_kernels.h — frustum culling and draw demultiplexing for gfx950.
//
// Reference semantics (file:line relative to the reference root):
//   frustum_culling + cull        shader/src/lib.rs:411-465     (dispatch: src/main.rs:1716-1763)
//   demultiplex_draws             shader/src/lib.rs:467-517     (dispatch: src/main.rs:1811-1838)
//   Similarity * Vec3, unpack     shared-structs/src/lib.rs:178-236
// Decisions are comparisons of fp32 values, so the arithmetic follows the compiled shaders' operation order with
// contraction off: the visible sets are identical to the CPU restatement's, not merely close.
#pragma once

#include "tr_common.h"

namespace tr {

// glam 0.19 scalar Quat * Vec3 (q = x, y, z, w): v (w^2 - b.b) + b (2 v.b) + (b x v) (2 w)
__device__ __forceinline__ void quat_rotate(const float q[4], float vx, float vy, float vz, float out[3]) {
#pragma clang fp contract(off)
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const float b2 = (qx * qx + qy * qy) + qz * qz;
    const float s0 = qw * qw - b2;
    const float s1 = ((vx * qx + vy * qy) + vz * qz) * 2.0f;
    const float s2 = qw * 2.0f;
    const float crx = qy * vz - vy * qz, cry = qz * vx - vz * qx, crz = qx * vy - vx * qy;
    out[0] = (vx * s0 + qx * s1) + crx * s2;
    out[1] = (vy * s0 + qy * s1) + cry * s2;
    out[2] = (vz * s0 + qz * s1) + crz * s2;
}

// Similarity * Vec3: translation + scale * (rotation * v)
__device__ __forceinline__ void similarity_apply(const tr_instance& inst, float vx, float vy, float vz, float out[3]) {
#pragma clang fp contract(off)
    float r[3];
    quat_rotate(inst.rotation, vx, vy, vz, r);
    const float s = inst.translation_and_scale[3];
    out[0] = inst.translation_and_scale[0] + r[0] * s;
    out[1] = inst.translation_and_scale[1] + r[1] * s;
    out[2] = inst.translation_and_scale[2] + r[2] * s;
}

__device__ __forceinline__ void mat4_mul_point(const float* m, float x, float y, float z, float out[4]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // glam 0.19 Mat4 * Vec4: ((X*x + Y*y) + Z*z) + W*1
        float acc = m[0 + r] * x;
        acc = m[4 + r] * y + acc;
        acc = m[8 + r] * z + acc;
        acc = m[12 + r] * 1.0f + acc;
        out[r] = acc;
    }
}

struct tr_cull_params {
    tr_culling_push_constants pc;
    uint32_t num_instances, num_primitives;
};

// One thread per instance; the only shared state is one counter per primitive, and only its final value is
// observable, so the atomic order does not matter.
__device__ __forceinline__ void frustum_culling_body(const tr_cull_params& p, const tr_primitive_info* __restrict__ primitives,
                                                     const tr_instance* __restrict__ instances,
                                                     uint32_t* __restrict__ instance_counts, uint32_t block) {
#pragma clang fp contract(off)
    const uint32_t i = block * 256u + threadIdx.x;
    bool visible = false;
    uint32_t primitive = 0u;
    if (i < p.num_instances) {
        const tr_instance inst = instances[i];
        primitive = inst.primitive_id;
        if (primitive < p.num_primitives) {   // unchecked in the reference
            const tr_primitive_info prim = primitives[primitive];
            float c[3], v[4];
            similarity_apply(inst, prim.packed_bounding_sphere[0], prim.packed_bounding_sphere[1], prim.packed_bounding_sphere[2], c);
            mat4_mul_point(p.pc.view, c[0], c[1], c[2], v);
            const float cx = v[0], cy = v[1], cz = -v[2];          // "in the view, +z = back so we flip it"
            const float radius = prim.packed_bounding_sphere[3] * inst.translation_and_scale[3];
            visible = cz + radius > p.pc.z_near;
            visible &= cz * p.pc.frustum_x_xz[1] - fabsf(cx) * p.pc.frustum_x_xz[0] < radius;
            visible &= cz * p.pc.frustum_y_yz[1] - fabsf(cy) * p.pc.frustum_y_yz[0] < radius;
        }
    }
    // One add per primitive and WAVE, not per instance: the instances of a primitive sit next to each other, and atomics on
    // one address complete 11-14 ns apart on this chip — 64 lanes adding to the same counter are a queue.
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t todo = __builtin_amdgcn_ballot_w64(visible);
    while (todo != 0ull) {
        const int first = __ffsll((unsigned long long)todo) - 1;
        const uint32_t pid = (uint32_t)__builtin_amdgcn_readlane((int)primitive, first);
        const uint64_t same = __builtin_amdgcn_ballot_w64(visible && primitive == pid);
        if ((int)lane == first) atomicAdd(&instance_counts[pid], (uint32_t)__popcll((unsigned long long)same));
        todo &= ~same;
    }
}
__global__ __launch_bounds__(256) void frustum_culling_kernel(const tr_cull_params p,
                                                              const tr_primitive_info* __restrict__ primitives,
                                                              const tr_instance* __restrict__ instances,
                                                              uint32_t* __restrict__ instance_counts) {
    frustum_culling_body(p, primitives, instances, instance_counts, blockIdx.x);
}

struct tr_draw_buffers {
    tr_draw_command* draws[TR_NUM_DRAW_BUFFERS];
};

// One workgroup of 1024 threads walks the primitives in chunks of 1024 and appends, per draw buffer, the commands of
// the primitives with a non-zero instance count IN ASCENDING PRIMITIVE ORDER (ballot + mbcnt inside a wave, LDS
// across the 16 waves): the reference's atomic append is order-free, this one is reproducible.  The work is a few
// bytes per primitive; one workgroup is latency-, not throughput-bound, up to ~1e6 primitives.
// ZERO_COUNTS (the frame recorder's own buffer): every count is zeroed once it has been read, so the next frame's
// culling finds the buffer clear without a fill launch of its own (src/main.rs:1668-1674 zeroes it per frame).
// NT: threads of the workgroup (1024, or 256 when the frame's first launch runs this behind its culling blocks: then the
// counts are read with agent-scope loads — other XCDs' culling blocks have just incremented them by atomics).
template <bool ZERO_COUNTS, uint32_t NT = 1024u, bool COUNTS_FROM_ATOMICS = false>
__device__ __forceinline__ void demultiplex_draws_body(const tr_primitive_info* __restrict__ primitives,
                                                       uint32_t* __restrict__ instance_counts, uint32_t num_primitives,
                                                       uint32_t* __restrict__ draw_counts, const tr_draw_buffers& out) {
    __shared__ uint32_t wave_totals[16][TR_NUM_DRAW_BUFFERS];
    __shared__ uint32_t running[TR_NUM_DRAW_BUFFERS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < TR_NUM_DRAW_BUFFERS) running[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < num_primitives; base += NT) {
        const uint32_t d = base + threadIdx.x;
        uint32_t n = 0, buffer = 0;
        tr_primitive_info prim;
        if (d < num_primitives) {
            if constexpr (COUNTS_FROM_ATOMICS) n = __hip_atomic_load(&instance_counts[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else n = instance_counts[d];
            if constexpr (ZERO_COUNTS) instance_counts[d] = 0u;
            prim = primitives[d];
            buffer = prim.draw_buffer_index < 3u ? prim.draw_buffer_index : 3u;   // the `_ =>` arm
        }
        uint32_t rank_in_wave = 0;
#pragma unroll
        for (uint32_t b = 0; b < TR_NUM_DRAW_BUFFERS; ++b) {
            const uint64_t mask = __ballot(n != 0u && buffer == b);
            if (buffer == b)
                rank_in_wave = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (lane == 0) wave_totals[wave][b] = (uint32_t)__popcll(mask);
        }
        __syncthreads();
        if (n != 0u) {
            uint32_t slot = running[buffer] + rank_in_wave;
            for (uint32_t w = 0; w < wave; ++w) slot += wave_totals[w][buffer];
            tr_draw_command c;
            c.index_count = prim.index_count;
            c.instance_count = n;
            c.first_index = prim.first_index;
            c.vertex_offset = 0;
            c.first_instance = prim.first_instance;
            out.draws[buffer][slot] = c;
        }
        __syncthreads();
        if (threadIdx.x < TR_NUM_DRAW_BUFFERS) {
            uint32_t t = 0;
            for (uint32_t w = 0; w < NT / 64u; ++w) t += wave_totals[w][threadIdx.x];
            running[threadIdx.x] += t;
        }
        __syncthreads();
    }
    if (threadIdx.x < TR_NUM_DRAW_BUFFERS) draw_counts[threadIdx.x] = running[threadIdx.x];
}
__global__ __launch_bounds__(1024) void demultiplex_draws_kernel(const tr_primitive_info* __restrict__ primitives,
                                                                 const uint32_t* __restrict__ instance_counts,
                                                                 uint32_t num_primitives,
                                                                 uint32_t* __restrict__ draw_counts,
                                                                 const tr_draw_buffers out) {
    demultiplex_draws_body<false>(primitives, const_cast<uint32_t*>(instance_counts), num_primitives, draw_counts, out);
}

}  // namespace tr
