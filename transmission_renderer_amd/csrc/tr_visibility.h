// tr_visibility.h — what the rasteriser leaves per pixel and per triangle, and how a pixel's attributes come out of it.
//
// A visibility word is (depth bits << 32 | triangle record index), 0 = no fragment.  The resolve (tr_raster_kernels.h)
// turns the words into TGB-v1 planes for the stand-alone passes; inside the frame recorder the shading kernels read the
// words themselves (shade_kernel<.., VIS = true>, tr_kernels.h) and the planes are never written.  Both interpolate a
// pixel's attributes from the triangle's plane equations (tr_tri_planes, vis_planes_interpolate); coverage and depth are
// the rasteriser's: the fp32 operations of oracle/tr_oracle.c `o_rasterize`, contraction off (tri_edges, tri_depth).
#pragma once

#include "tr_common.h"

namespace tr {

// What the RASTERISER reads of a triangle (wave-uniform there: scalar loads): one 128-byte line.
struct alignas(16) tr_tri_record {
    float A[3], B[3], C[3];   // edge functions (positive inside)
    float z[3], w[3];         // clip z, w per vertex
    uint16_t x0, y0, x1, y1;  // inclusive pixel bounds (x0 > x1: culled / empty)
    uint32_t flags;           // bit0: alpha clipped draw; bits 1-2: the material's class for the tile coverage words
                              // (2 = full-class textured, 4 = anything else, 6 = not known yet)
    uint32_t material_id;     // of the instance
    float T[3][2];            // uv per corner (the alpha-clip kill samples the base colour)
    uint32_t _pad[7];
};
static_assert(sizeof(tr_tri_record) == 128, "tr_tri_record is one 128 B line");

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
        // (bitwise, not short-circuit: two compares and two mask operations per edge; the short-circuit form is compiled to a
        //  compare under a saved exec mask per edge — six instructions, three of them exec writes)
        const bool positive = fv[i] > 0.0f, on_edge = fv[i] == 0.0f;
        inside = inside & (positive | (on_edge & tie));
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


// What a SHADING launch (or the resolve) needs of a triangle: perspective-correct interpolation as plane equations.
// With the homogeneous edge functions f_i(x, y) = A_i x + B_i y + C_i of the record, the attribute of the pixel is
//     a(x, y) = (f_0 a_0 + f_1 a_1 + f_2 a_2) / (f_0 + f_1 + f_2)
// and numerator and denominator are themselves planes in (x, y): three coefficients each, formed once per triangle by the
// set-up (in fp64 from the record's fp32 edge functions — the ones that decided coverage — and re-centred on the
// triangle's bounds origin, so the per-pixel evaluation has no large-offset cancellation).  A pixel costs two
// multiply-adds per plane, one v_rcp_f32 and one multiply per attribute: ~35 vector instructions instead of the ~85 of
// barycentrics + IEEE division + per-corner mixes, and the record is one 128-byte line instead of 192 bytes.
// Coverage, the depth winner, depth and ids are NOT touched by this (they come from the rasteriser's words: bit-exact
// against the oracle); position / normal / uv agree with the oracle's own fp32 evaluation to a few 1e-7 relative — its
// rounding noise, not this form's (tests/test_gpu_raster.py states the bound).
struct alignas(16) tr_tri_planes {
    float den[3];         // f_0 + f_1 + f_2 at (origin + (dx, dy) + 0.5): den[0] dx + den[1] dy + den[2]
    uint32_t origin;      // x0 | y0 << 16: the pixel the planes are centred on (the triangle's clamped bounds corner)
    float attr[8][3];     // numerator planes: position xyz, normal xyz, uv
    uint32_t material_id; // of the instance
    float scale;          // of the instance (translation_and_scale.w)
    uint32_t _pad[2];
};
static_assert(sizeof(tr_tri_planes) == 128, "tr_tri_planes is one 128 B line");

// The winning triangle's attributes at the pixel centre (vertex_instanced_with_scale outputs, perspective-correct).
struct vis_fragment {
    float position[3], depth;
    float normal[3], scale;
    float uv[2];
    uint32_t material_id;
};
// A pixel's attributes from its winning triangle's planes.
template <class Planes>
__device__ __forceinline__ void vis_planes_interpolate(const Planes& R, unsigned long long key, uint32_t px, uint32_t py, vis_fragment& o) {
    const uint32_t origin = R.origin;
    const float dx = (float)((int)px - (int)(origin & 0xFFFFu)), dy = (float)((int)py - (int)(origin >> 16));
    auto plane = [&](const float (&c)[3]) { return fmaf(c[0], dx, fmaf(c[1], dy, c[2])); };
    const float inv = __builtin_amdgcn_rcpf(plane(R.den));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.position[k] = plane(R.attr[k]) * inv;
        o.normal[k] = plane(R.attr[3 + k]) * inv;
    }
    o.uv[0] = plane(R.attr[6]) * inv;
    o.uv[1] = plane(R.attr[7]) * inv;
    o.depth = __uint_as_float((uint32_t)(key >> 32));
    o.scale = R.scale;
    o.material_id = R.material_id;
}

// Set-up side: the planes of one triangle from its record and the vertex stage's outputs per corner (world position,
// rotated normal; uv is in the record) — fp64: once per triangle, not per pixel.
__device__ __forceinline__ void tri_planes_from_record(const tr_tri_record& r, const float P[3][3], const float N[3][3], float scale,
                                                       tr_tri_planes& out) {
    const bool valid = r.x0 <= r.x1;
    const uint32_t x0 = valid ? r.x0 : 0u, y0 = valid ? r.y0 : 0u;
    const double ox = (double)x0 + 0.5, oy = (double)y0 + 0.5;
    const double A[3] = {r.A[0], r.A[1], r.A[2]}, B[3] = {r.B[0], r.B[1], r.B[2]}, C[3] = {r.C[0], r.C[1], r.C[2]};
    auto plane = [&](double a0, double a1, double a2, float (&c)[3]) {
        const double pa = (A[0] * a0 + A[1] * a1) + A[2] * a2, pb = (B[0] * a0 + B[1] * a1) + B[2] * a2;
        const double pc = (C[0] * a0 + C[1] * a1) + C[2] * a2;
        c[0] = (float)pa;
        c[1] = (float)pb;
        c[2] = (float)((pa * ox + pb * oy) + pc);
    };
    plane(1.0, 1.0, 1.0, out.den);
    for (int k = 0; k < 3; ++k) {
        plane(P[0][k], P[1][k], P[2][k], out.attr[k]);
        plane(N[0][k], N[1][k], N[2][k], out.attr[3 + k]);
    }
    plane(r.T[0][0], r.T[1][0], r.T[2][0], out.attr[6]);
    plane(r.T[0][1], r.T[1][1], r.T[2][1], out.attr[7]);
    out.origin = x0 | (y0 << 16);
    out.material_id = r.material_id;
    out.scale = scale;
    out._pad[0] = out._pad[1] = 0u;
}

}  // namespace tr
