// tr_visibility.h — what the rasteriser leaves per pixel and per triangle, and how a pixel's attributes come out of it.
//
// A visibility word is (depth bits << 32 | triangle record index), 0 = no fragment.  The resolve (tr_raster_kernels.h)
// turns the words into TGB-v1 planes for the stand-alone passes; inside the frame recorder the shading kernels read the
// words themselves (shade_kernel<.., VIS = true>, tr_kernels.h) and the planes are never written.  Both call
// vis_interpolate: same fp32 operations, contraction off, as oracle/tr_oracle.c `o_rasterize`.
#pragma once

#include "tr_common.h"

namespace tr {

struct alignas(16) tr_tri_record {
    float A[3], B[3], C[3];   // edge functions (positive inside)
    float z[3], w[3];         // clip z, w per vertex
    uint16_t x0, y0, x1, y1;  // inclusive pixel bounds (x0 > x1: culled / empty)
    uint32_t v[3];            // vertex indices
    uint32_t instance;
    uint32_t flags;           // bit0: alpha clipped draw; bits 1-2: the material's class for the tile coverage words
                              // (2 = full-class textured, 4 = anything else, 6 = not known yet)
    uint32_t material_id;     // of the instance
    float scale;              // of the instance (translation_and_scale.w)
    // the vertex stage's outputs per corner (vertex_instanced_with_scale: world position, rotated normal, uv), so the
    // resolve interpolates without redoing three vertex stages per PIXEL
    float P[3][3], N[3][3], T[3][2];
};
static_assert(sizeof(tr_tri_record) == 192, "tr_tri_record is 192 B");

// Barycentrics and depth at a pixel centre (oracle: tri_pixel).  `rec` is wave-uniform in the raster kernel
// (scalar registers) and per lane in the resolve.
// The three edge functions at a pixel centre and the top-left-rule inside test.
template <class Rec>
__device__ __forceinline__ bool tri_edges(const Rec& rec, float pxc, float pyc, float fv[3]) {
#pragma clang fp contract(off)
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fv[i] = (rec.A[i] * pxc + rec.B[i] * pyc) + rec.C[i];
        const bool tie = rec.A[i] > 0.0f || (rec.A[i] == 0.0f && rec.B[i] > 0.0f);
        inside &= fv[i] > 0.0f || (fv[i] == 0.0f && tie);
    }
    return inside;
}

// Barycentrics and depth from the edge values; true if the fragment survives clipping and the depth range.
template <class Rec>
__device__ __forceinline__ bool tri_depth(const Rec& rec, const float fv[3], float lambda[3], float& depth) {
#pragma clang fp contract(off)
    const float sum = (fv[0] + fv[1]) + fv[2];
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 3; ++i) lambda[i] = fv[i] * inv;
    const float zc = (lambda[0] * rec.z[0] + lambda[1] * rec.z[1]) + lambda[2] * rec.z[2];
    const float wc = (lambda[0] * rec.w[0] + lambda[1] * rec.w[1]) + lambda[2] * rec.w[2];
    depth = zc / wc;
    return sum > 0.0f && wc > 0.0f && zc <= wc && depth > 0.0f;
}

template <class Rec>
__device__ __forceinline__ bool tri_pixel(const Rec& rec, float pxc, float pyc, float lambda[3], float& depth) {
    float fv[3];
    const bool inside = tri_edges(rec, pxc, pyc, fv);
    return tri_depth(rec, fv, lambda, depth) && inside;
}


// The winning triangle's attributes at the pixel centre (vertex_instanced_with_scale outputs, perspective-correct).
struct vis_fragment {
    float position[3], depth;
    float normal[3], scale;
    float uv[2];
    uint32_t material_id;
};
template <class Rec>
__device__ __forceinline__ void vis_interpolate(const Rec& rec, unsigned long long key, uint32_t px, uint32_t py, vis_fragment& o) {
#pragma clang fp contract(off)
    float lam[3], depth;
    tri_pixel(rec, (float)px + 0.5f, (float)py + 0.5f, lam, depth);
    auto mix = [&](float a, float b, float c) { return (lam[0] * a + lam[1] * b) + lam[2] * c; };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.position[k] = mix(rec.P[0][k], rec.P[1][k], rec.P[2][k]);
        o.normal[k] = mix(rec.N[0][k], rec.N[1][k], rec.N[2][k]);
    }
    o.depth = __uint_as_float((uint32_t)(key >> 32));
    o.scale = rec.scale;
    o.uv[0] = mix(rec.T[0][0], rec.T[1][0], rec.T[2][0]);
    o.uv[1] = mix(rec.T[0][1], rec.T[1][1], rec.T[2][1]);
    o.material_id = rec.material_id;
}

}  // namespace tr
