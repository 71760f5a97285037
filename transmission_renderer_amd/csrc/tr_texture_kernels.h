// tr_texture_kernels.h — the bindless material textures: upload-time mip chain and the
// per-pixel sampler the textured shading kernels call.
//
// Reference semantics (file:line relative to the reference root):
//   load_image_from_bytes: full mip chain by LINEAR blits      src/model_loading.rs:335-390
//   `sampler` (LINEAR min/mag/mip, REPEAT, anisotropy off)     src/main.rs:683-692
//   TextureSampler::sample -> OpImageSampleImplicitLod         shader/src/lib.rs:252-262
// Storage: one arena in HBM holding every texture's chain, RGBA8, levels packed; tr_dtex (scalar-loaded: the
// texture id is a property of the material, and a wave shades one material at a time) describes one chain.
#pragma once

#include "tr_common.h"

namespace tr {

struct alignas(16) tr_dtex {
    uint32_t width, height, levels, srgb;
    float wf, hf, max_lod;
    uint32_t _pad;
    // texel offset of each level from the arena start; entries [levels ..] repeat the last level, so that the pair
    // (offset[l], offset[l+1]) read at the per-pixel level l is always the two levels the LINEAR mip filter blends
    uint32_t offset[TR_MAX_MIP_LEVELS + 4];
};
static_assert(sizeof(tr_dtex) == 112, "tr_dtex is 112 B");
typedef const TR_CONSTANT tr_dtex cdtex;

// Decode / encode tables shared by the mip builder and the sampler; filled on the host with the libm the CPU
// restatement links, so the device's R8G8B8A8_SRGB decode is that function exactly.
struct tr_colour_tables {
    float srgb_to_linear[256];   // Khronos data format spec 13.3
    float unorm[256];            // b / 255
    float srgb_threshold[256];   // smallest linear value that encodes to >= b (entry 0 = 0)
};

typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// ------------------------------------------------------------------------ upload-time mip chain
// Level l from level l-1 as a LINEAR blit of the whole image (bilinear about the destination texel centre, clamp to
// edge), fp32 in the oracle's operation order with no contraction: the chain is byte-identical to
// oracle/tr_oracle.c o_generate_texture_mips.  One thread per destination texel.
__global__ __launch_bounds__(256) void texture_downsample_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                                 uint32_t ws, uint32_t hs, uint32_t wd, uint32_t hd,
                                                                 uint32_t srgb, const tr_colour_tables* __restrict__ tab) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t j = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (i >= wd || j >= hd) return;
    const float sx = (float)ws / (float)wd, sy = (float)hs / (float)hd;
    const float y = ((float)j + 0.5f) * sy - 0.5f;
    const float fy0 = floorf(y), by = y - fy0;
    int y0 = (int)fy0, y1 = y0 + 1;
    y0 = min(max(y0, 0), (int)hs - 1);
    y1 = min(y1, (int)hs - 1);
    const float x = ((float)i + 0.5f) * sx - 0.5f;
    const float fx0 = floorf(x), ax = x - fx0;
    int x0 = (int)fx0, x1 = x0 + 1;
    x0 = min(max(x0, 0), (int)ws - 1);
    x1 = min(x1, (int)ws - 1);
    const float w00 = (1.0f - ax) * (1.0f - by), w10 = ax * (1.0f - by), w01 = (1.0f - ax) * by, w11 = ax * by;
    const uint32_t t00 = src[(size_t)y0 * ws + (uint32_t)x0], t10 = src[(size_t)y0 * ws + (uint32_t)x1];
    const uint32_t t01 = src[(size_t)y1 * ws + (uint32_t)x0], t11 = src[(size_t)y1 * ws + (uint32_t)x1];
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* dec = (srgb && k < 3) ? tab->srgb_to_linear : tab->unorm;
        const float a = dec[(t00 >> (8 * k)) & 0xFFu], b = dec[(t10 >> (8 * k)) & 0xFFu];
        const float c = dec[(t01 >> (8 * k)) & 0xFFu], d = dec[(t11 >> (8 * k)) & 0xFFu];
        float r = (a * w00 + b * w10) + (c * w01 + d * w11);
        if (!(r > 0.0f)) r = 0.0f;
        if (r > 1.0f) r = 1.0f;
        uint32_t byte;
        if (srgb && k < 3) {
            uint32_t lo = 0, hi = 255;   // largest b with srgb_threshold[b] <= r
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (r >= tab->srgb_threshold[mid]) lo = mid;
                else hi = mid - 1u;
            }
            byte = lo;
        } else {
            byte = (uint32_t)(r * 255.0f + 0.5f);
        }
        out |= byte << (8 * k);
    }
    dst[(size_t)j * wd + i] = out;
}

// ------------------------------------------------------------------------ the sampler
// Screen-space differences of uv inside the pixel's 2x2 quad (OpDPdx / OpDPdy of the interpolant).
struct uv_derivs {
    float dudx, dvdx, dudy, dvdy;
};

struct texel_quad {   // the four taps of one level and their weights
    uint32_t t00, t10, t01, t11;
    float fx, fy;
};

__device__ __forceinline__ void texture_issue_level(texel_quad& q, const uint32_t* __restrict__ level_base, uint32_t w,
                                                    uint32_t h, float uu, float vv) {
    // REPEAT: the coordinate is wrapped to [0,1) by the caller; x = u*w - 0.5, taps floor(x), floor(x)+1 modulo w
    const float x = fmaf(uu, (float)w, -0.5f), y = fmaf(vv, (float)h, -0.5f);
    const float fx0 = floorf(x), fy0 = floorf(y);
    q.fx = x - fx0;
    q.fy = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 += x0 < 0 ? (int)w : 0;
    y0 += y0 < 0 ? (int)h : 0;
    x1 -= x1 >= (int)w ? (int)w : 0;
    y1 -= y1 >= (int)h ? (int)h : 0;
    // non-finite coordinates only: keep the addresses inside the level (the result is NaN anyway)
    const uint32_t ux0 = min((uint32_t)x0, w - 1u), uy0 = min((uint32_t)y0, h - 1u);
    const uint32_t ux1 = min((uint32_t)x1, w - 1u), uy1 = min((uint32_t)y1, h - 1u);
    q.t00 = level_base[uy0 * w + ux0];
    q.t10 = level_base[uy0 * w + ux1];
    q.t01 = level_base[uy1 * w + ux0];
    q.t11 = level_base[uy1 * w + ux1];
}

struct texture_fetch {
    texel_quad q[2];
    float frac;
};

// Issues the eight taps of texture.sample(sampler, uv) with implicit LOD (Vulkan 1.3 "Scale Factor Operation"):
//   rho = max(|(du/dx w, dv/dx h)|, |(du/dy w, dv/dy h)|), lambda = log2(rho) clamped to [0, levels-1],
//   LINEAR between floor(lambda) and the next level.
// (`head`: the descriptor's scalars, which a caller with several fetches of one texture reads once — the rasteriser's alpha kill)
struct tr_dtex_head {
    uint32_t width, height, levels;
    float wf, hf, max_lod;
};
__device__ __forceinline__ tr_dtex_head texture_head(cdtex* t) { return {t->width, t->height, t->levels, t->wf, t->hf, t->max_lod}; }
__device__ __forceinline__ void texture_issue(texture_fetch& f, const uint32_t* __restrict__ arena, const tr_dtex_head& t,
                                              const TR_CONSTANT uint32_t* level_offset, float u, float v, const uv_derivs& d) {
    const float wf = t.wf, hf = t.hf;
    const float mxx = d.dudx * wf, mxy = d.dvdx * hf, myx = d.dudy * wf, myy = d.dvdy * hf;
    const float rho2 = fmaxf(fmaf(mxx, mxx, mxy * mxy), fmaf(myx, myx, myy * myy));
    const float lambda = 0.5f * fast_log2(rho2);                   // log2(sqrt(rho2)); rho2 = 0 -> -inf -> level 0
    const float l = fminf(fmaxf(lambda, 0.0f), t.max_lod);        // NaN -> 0
    const float lf = floorf(l);
    f.frac = l - lf;
    const uint32_t l0 = (uint32_t)lf;
    const uint32_t l1 = min(l0 + 1u, t.levels - 1u);
    // the level offsets are the only per-pixel table read: one 8-byte load of (offset[l0], offset[l0 + 1])
    const u32x2_a4 o = *reinterpret_cast<const TR_CONSTANT u32x2_a4*>(level_offset + l0);
    const uint32_t w = t.width, h = t.height;
    const float uu = u - floorf(u), vv = v - floorf(v);
    texture_issue_level(f.q[0], arena + o.x, max(w >> l0, 1u), max(h >> l0, 1u), uu, vv);
    texture_issue_level(f.q[1], arena + o.y, max(w >> l1, 1u), max(h >> l1, 1u), uu, vv);
}
__device__ __forceinline__ void texture_issue(texture_fetch& f, const uint32_t* __restrict__ arena, cdtex* t, float u,
                                              float v, const uv_derivs& d) {
    texture_issue(f, arena, texture_head(t), t->offset, u, v, d);
}

// Filters channel `k` of the fetched taps.  sRGB channels are decoded through the LDS copy of the table before
// filtering (the filter runs in linear light); UNORM channels are filtered as byte values and scaled once.
template <int K>
__device__ __forceinline__ float texture_resolve_channel(const texture_fetch& f, bool srgb, const float* __restrict__ lds_srgb) {
    float lv[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const texel_quad& q = f.q[l];
        float a, b, c, d;
        if (srgb && K < 3) {
            a = lds_srgb[(q.t00 >> (8 * K)) & 0xFFu];
            b = lds_srgb[(q.t10 >> (8 * K)) & 0xFFu];
            c = lds_srgb[(q.t01 >> (8 * K)) & 0xFFu];
            d = lds_srgb[(q.t11 >> (8 * K)) & 0xFFu];
        } else {
            a = (float)((q.t00 >> (8 * K)) & 0xFFu);   // v_cvt_f32_ubyteK
            b = (float)((q.t10 >> (8 * K)) & 0xFFu);
            c = (float)((q.t01 >> (8 * K)) & 0xFFu);
            d = (float)((q.t11 >> (8 * K)) & 0xFFu);
        }
        const float top = fmaf(b - a, q.fx, a), bot = fmaf(d - c, q.fx, c);
        lv[l] = fmaf(bot - top, q.fy, top);
    }
    const float r = fmaf(lv[1] - lv[0], f.frac, lv[0]);
    return (srgb && K < 3) ? r : r * (1.0f / 255.0f);
}

// ---- the same sampler with the geometry factored out: materials usually bind several textures of one size, and
// everything up to the tap addresses (LOD, level pair, wrapped tap coordinates, weights) depends on the size only.
struct tex_geom {
    uint32_t o[2][4];      // byte offsets of the taps t00, t10, t01, t11 from the chain's first texel, per level
    float fx[2], fy[2], frac;
};
struct tex_taps {
    uint32_t t[2][4];
};

__device__ __forceinline__ void tex_geom_level(tex_geom& g, int lv, uint32_t rel, uint32_t w, uint32_t h, float uu, float vv) {
    const float x = fmaf(uu, (float)w, -0.5f), y = fmaf(vv, (float)h, -0.5f);
    const float fx0 = floorf(x), fy0 = floorf(y);
    g.fx[lv] = x - fx0;
    g.fy[lv] = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 += x0 < 0 ? (int)w : 0;
    y0 += y0 < 0 ? (int)h : 0;
    x1 -= x1 >= (int)w ? (int)w : 0;
    y1 -= y1 >= (int)h ? (int)h : 0;
    const uint32_t ux0 = min((uint32_t)x0, w - 1u), uy0 = min((uint32_t)y0, h - 1u);
    const uint32_t ux1 = min((uint32_t)x1, w - 1u), uy1 = min((uint32_t)y1, h - 1u);
    const uint32_t r0 = mad24(uy0, w, rel), r1 = mad24(uy1, w, rel);
    g.o[lv][0] = (r0 + ux0) * 4u;
    g.o[lv][1] = (r0 + ux1) * 4u;
    g.o[lv][2] = (r1 + ux0) * 4u;
    g.o[lv][3] = (r1 + ux1) * 4u;
}

__device__ __forceinline__ void tex_geom_compute(tex_geom& g, cdtex* t, float u, float v, const uv_derivs& d) {
    const float wf = t->wf, hf = t->hf;
    const float mxx = d.dudx * wf, mxy = d.dvdx * hf, myx = d.dudy * wf, myy = d.dvdy * hf;
    const float rho2 = fmaxf(fmaf(mxx, mxx, mxy * mxy), fmaf(myx, myx, myy * myy));
    const float lambda = 0.5f * fast_log2(rho2);
    const float l = fminf(fmaxf(lambda, 0.0f), t->max_lod);
    const float lf = floorf(l);
    g.frac = l - lf;
    const uint32_t l0 = (uint32_t)lf;
    const uint32_t l1 = min(l0 + 1u, t->levels - 1u);
    const u32x2_a4 o = *reinterpret_cast<const TR_CONSTANT u32x2_a4*>(t->offset + l0);
    const uint32_t base = t->offset[0];                 // (scalar) the same for every texture of this size: relative
    const uint32_t w = t->width, h = t->height;
    const float uu = u - floorf(u), vv = v - floorf(v);
    tex_geom_level(g, 0, o.x - base, max(w >> l0, 1u), max(h >> l0, 1u), uu, vv);
    tex_geom_level(g, 1, o.y - base, max(w >> l1, 1u), max(h >> l1, 1u), uu, vv);
}

template <int LEVELS = 2>
__device__ __forceinline__ void texture_issue_shared(tex_taps& f, const uint32_t* __restrict__ arena, cdtex* t, const tex_geom& g) {
    const uint32_t* chain = arena + t->offset[0];       // scalar base: the taps are saddr + voffset loads
#pragma unroll
    for (int lv = 0; lv < LEVELS; ++lv)
#pragma unroll
        for (int k = 0; k < 4; ++k) f.t[lv][k] = ld<uint32_t>(chain, g.o[lv][k]);
}

// LEVELS = 1: the wave's pixels all sit exactly on their lower level (frac == 0: a magnified texture, lambda clamped to
// 0): the upper level's taps would be multiplied by zero — (lv1 - lv0) * 0 + lv0 == lv0 for finite texels — and are
// neither fetched nor decoded.
template <int K, int LEVELS = 2>
__device__ __forceinline__ float texture_resolve_shared(const tex_taps& f, const tex_geom& g, bool srgb,
                                                        const float* __restrict__ lds_srgb) {
    float lv[2];
#pragma unroll
    for (int l = 0; l < LEVELS; ++l) {
        float a, b, c, d;
        if (srgb && K < 3) {
            a = lds_srgb[(f.t[l][0] >> (8 * K)) & 0xFFu];
            b = lds_srgb[(f.t[l][1] >> (8 * K)) & 0xFFu];
            c = lds_srgb[(f.t[l][2] >> (8 * K)) & 0xFFu];
            d = lds_srgb[(f.t[l][3] >> (8 * K)) & 0xFFu];
        } else {
            a = (float)((f.t[l][0] >> (8 * K)) & 0xFFu);
            b = (float)((f.t[l][1] >> (8 * K)) & 0xFFu);
            c = (float)((f.t[l][2] >> (8 * K)) & 0xFFu);
            d = (float)((f.t[l][3] >> (8 * K)) & 0xFFu);
        }
        const float top = fmaf(b - a, g.fx[l], a), bot = fmaf(d - c, g.fx[l], c);
        lv[l] = fmaf(bot - top, g.fy[l], top);
    }
    const float r = LEVELS == 2 ? fmaf(lv[1] - lv[0], g.frac, lv[0]) : lv[0];
    return (srgb && K < 3) ? r : r * (1.0f / 255.0f);
}

}  // namespace tr
