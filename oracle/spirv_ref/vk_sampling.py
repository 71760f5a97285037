"""Vulkan fixed-function image operations in numpy fp32, written from the Vulkan 1.3 specification — NOT from
oracle/tr_oracle.c.  TEST INFRASTRUCTURE: tools/make_golden_spirv.py answers the SPIR-V interpreter's OpImageSample*
callbacks and builds the mip chains with THIS module, so that the committed fixtures are "the reference's compiled
shader binary + the Vulkan specification" and contain no output of the C oracle; tests/test_vk_sampling.py then checks
that the oracle's samplers agree with it bit for bit.

What the specification fixes, and what this module has to choose (stated once, here):

* Vulkan 1.3 §16.5.8 "Scale Factor Operation" / §16.5.9 "LOD Operation and Image Level(s) Selection": for implicit-LOD
  sampling  m_ux = du/dx * w, m_vx = dv/dx * h, ...;  rho_x = sqrt(m_ux^2 + m_vx^2), rho_y likewise;
  rho_max = max(rho_x, rho_y);  lambda_base = log2(rho_max / eta), eta = 1 without anisotropy; lambda' = lambda_base +
  bias (0 here);  lambda = clamp(lambda', lod_min, min(lod_max, levels - 1)).  For VK_SAMPLER_MIPMAP_MODE_LINEAR:
  d_hi = floor(lambda), d_lo = min(d_hi + 1, q), delta = frac(lambda).  (The specification lets an implementation
  approximate rho; the exact Euclidean form above is its reference definition and the one used.)
* §16.6 "Normalized texel coordinate operations", §16.7 "(u,v,w,a) to (i,j,k,l,n) Transformation and Array Layer
  Selection": u = s * width_level; for VK_FILTER_LINEAR  i0 = floor(u - 0.5), i1 = i0 + 1, alpha = frac(u - 0.5).
  §16.6.? "Wrapping operation": REPEAT  i = i mod size;  CLAMP_TO_EDGE  i = clamp(i, 0, size - 1).
* §16.8 "Texel Filtering": tau_2D = (1-a)(1-b) t_i0j0 + a(1-b) t_i1j0 + (1-a) b t_i0j1 + a b t_i1j1 and
  tau = (1 - delta) tau[d_hi] + delta tau[d_lo].  The specification gives the VALUE, not the order of the fp32
  operations nor the weight precision (>= 8 bit of subtexel precision is allowed).  CHOICE: exact fp32 weights, the
  interpolation evaluated in lerp form  t0 + (t1 - t0) * a  horizontally, then vertically, then between levels.
* §19.5 "Image Copies with Scaling" (vkCmdBlitImage, what a `generate_mips` helper records per level — the reference's
  helper lives in the un-vendored ash-opinionated-abstractions crate, so "a LINEAR blit chain, level l from level l-1,
  whole-image regions" is itself a restatement): for destination texel (i, j):  u_base = i + 0.5;
  u_offset = u_base - x_dst0;  u_scaled = u_offset * scale_u, scale_u = (x_src1 - x_src0) / (x_dst1 - x_dst0);
  u = u_scaled + x_src0;  then unnormalised-coordinate LINEAR filtering of the source level with CLAMP_TO_EDGE.
  CHOICE: the four weights (1-a)(1-b), a(1-b), (1-a)b, ab formed first, the sum evaluated as
  (t00 w00 + t10 w10) + (t01 w01 + t11 w11) in fp32, one rounding per operation, no fused multiply-add.
* §3.? "Floating-Point Format Conversions" / Khronos Data Format Specification 1.3 §13.3 (sRGB EOTF: x <= 0.04045 ?
  x / 12.92 : ((x + 0.055) / 1.055)^2.4; inverse: x <= 0.0031308 ? 12.92 x : 1.055 x^(1/2.4) - 0.055), §10.1
  (16-bit float).  Conversion of a filtered fp32 value to R16G16B16A16_SFLOAT: the rounding mode is implementation
  defined (round to nearest even or toward zero).  CHOICE: round to nearest even.  UNORM8 store: round to nearest
  (floor(x * 255 + 0.5)), required by "Conversion from floating-point to normalized fixed-point".
* pow / log2 go through the host's glibc powf / log2f (ctypes), the same library the reference's CPU-side Rust
  (`f32::powf`, `f32::log2` -> the platform libm on Linux) would call; sqrt is numpy's correctly rounded fp32 sqrt.

Scalar entry points (one sample per call, numpy float32 scalars: one IEEE rounding per operation) serve the interpreter;
the chain builders are vectorised over the image with the same per-element operations.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

f32 = np.float32

_libm = C.CDLL("libm.so.6")
_libm.powf.restype = C.c_float
_libm.powf.argtypes = [C.c_float, C.c_float]
_libm.log2f.restype = C.c_float
_libm.log2f.argtypes = [C.c_float]


def mip_levels_for_size(width: int, height: int) -> int:
    """floor(log2(min(w, h))) + 1: the level count the reference allocates (src/main.rs:2590-2592) — an allocation
    decision of the reference's host code, not a Vulkan rule."""
    return int(min(width, height)).bit_length()


def level_dim(d: int, level: int) -> int:
    """Vulkan §12.3.? "Image Miplevel Sizing": max(d >> level, 1)."""
    return max(d >> level, 1)


# ------------------------------------------------------------------------------------------------ texel decode
def srgb8_to_linear_table() -> np.ndarray:
    """KDFS 13.3 EOTF of every 8-bit code, fp32 (x = c / 255 in fp32; powf of the host libm)."""
    t = np.zeros(256, dtype=np.float32)
    for c in range(256):
        x = f32(c) / f32(255.0)
        if x <= f32(0.04045):
            t[c] = x / f32(12.92)
        else:
            t[c] = f32(_libm.powf(float((x + f32(0.055)) / f32(1.055)), float(f32(2.4))))
    return t


_SRGB_TABLE = None


def _srgb_table() -> np.ndarray:
    global _SRGB_TABLE
    if _SRGB_TABLE is None:
        _SRGB_TABLE = srgb8_to_linear_table()
    return _SRGB_TABLE


def decode_rgba8(texels: np.ndarray, srgb: bool) -> np.ndarray:
    """(..., 4) uint8 -> fp32: UNORM c / 255; sRGB formats decode R, G, B through the EOTF, alpha stays UNORM."""
    texels = np.asarray(texels, dtype=np.uint8)
    out = texels.astype(np.float32) / f32(255.0)
    if srgb:
        out[..., :3] = _srgb_table()[texels[..., :3]]
    return out


def linear_to_srgb8(x: np.ndarray) -> np.ndarray:
    """KDFS 13.3 inverse EOTF, then UNORM8 round to nearest; input clamped to [0, 1] (NaN -> 0)."""
    x = np.asarray(x, dtype=np.float32)
    x = np.where(x > f32(0.0), x, f32(0.0)).astype(np.float32)
    x = np.where(x > f32(1.0), f32(1.0), x).astype(np.float32)
    flat = x.reshape(-1)
    e = np.empty_like(flat)
    inv = f32(1.0) / f32(2.4)
    for i, v in enumerate(flat):
        if v <= f32(0.0031308):
            e[i] = f32(12.92) * v
        else:
            e[i] = f32(1.055) * f32(_libm.powf(float(v), float(inv))) - f32(0.055)
    return (e * f32(255.0) + f32(0.5)).astype(np.uint8).reshape(x.shape)


def unorm8_store(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    x = np.where(x > f32(0.0), x, f32(0.0)).astype(np.float32)
    x = np.where(x > f32(1.0), f32(1.0), x).astype(np.float32)
    return (x * f32(255.0) + f32(0.5)).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ blit chains (§19.5)
def _blit_axis(dst: int, src: int):
    """Per destination index: (i0, i1, alpha) of the LINEAR filter along one axis, whole-image regions, clamp to edge."""
    scale = f32(src) / f32(dst)
    u = (np.arange(dst, dtype=np.float32) + f32(0.5)) * scale - f32(0.5)    # u - 0.5 of §16.7
    fl = np.floor(u)
    alpha = (u - fl).astype(np.float32)
    i0 = fl.astype(np.int64)
    i1 = i0 + 1
    i0 = np.clip(i0, 0, src - 1)
    i1 = np.minimum(i1, src - 1)
    return i0, i1, alpha


def _blit_level_f32(src: np.ndarray, wd: int, hd: int) -> np.ndarray:
    """One vkCmdBlitImage VK_FILTER_LINEAR of a decoded (hs, ws, C) fp32 level to (hd, wd, C), fp32 result."""
    hs, ws = src.shape[:2]
    x0, x1, ax = _blit_axis(wd, ws)
    y0, y1, by = _blit_axis(hd, hs)
    one = f32(1.0)
    w00 = ((one - ax)[None, :] * (one - by)[:, None]).astype(np.float32)
    w10 = (ax[None, :] * (one - by)[:, None]).astype(np.float32)
    w01 = ((one - ax)[None, :] * by[:, None]).astype(np.float32)
    w11 = (ax[None, :] * by[:, None]).astype(np.float32)
    t00 = src[y0[:, None], x0[None, :]]
    t10 = src[y0[:, None], x1[None, :]]
    t01 = src[y1[:, None], x0[None, :]]
    t11 = src[y1[:, None], x1[None, :]]
    top = (t00 * w00[..., None]).astype(np.float32) + (t10 * w10[..., None]).astype(np.float32)
    bot = (t01 * w01[..., None]).astype(np.float32) + (t11 * w11[..., None]).astype(np.float32)
    return (top.astype(np.float32) + bot.astype(np.float32)).astype(np.float32)


def blit_chain_rgba16f(mip0: np.ndarray, levels: int = 0) -> List[np.ndarray]:
    """The opaque framebuffer's mip chain: level l = LINEAR blit of level l-1 (its ROUNDED half texels), RTNE store."""
    mip0 = np.ascontiguousarray(mip0, dtype=np.float16)
    h, w = mip0.shape[:2]
    n = levels or mip_levels_for_size(w, h)
    out = [mip0]
    for l in range(1, n):
        src = out[-1].astype(np.float32)
        out.append(_blit_level_f32(src, level_dim(w, l), level_dim(h, l)).astype(np.float16))
    return out


def blit_chain_rgba8(img: np.ndarray, srgb: bool) -> List[np.ndarray]:
    """A material texture's chain: sRGB images are filtered in linear light and re-encoded; UNORM rounds to nearest."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    out = [img]
    for l in range(1, mip_levels_for_size(w, h)):
        r = _blit_level_f32(decode_rgba8(out[-1], srgb), level_dim(w, l), level_dim(h, l))
        b = unorm8_store(r)
        if srgb:
            b[..., :3] = linear_to_srgb8(r[..., :3])
        out.append(b)
    return out


# ------------------------------------------------------------------------------------------------ sampling (§16.5-16.8)
def _linear_taps_clamp(coord, size: int):
    """§16.7 VK_FILTER_LINEAR along one axis with CLAMP_TO_EDGE: (i0, i1, alpha).  A non-finite coordinate (undefined
    in the specification) is taken through the clamp the way min/max propagate the finite operand."""
    x = f32(coord) * f32(size) - f32(0.5)
    x = f32(min(max(x, f32(-1.0)), f32(size))) if x == x else f32(-1.0)
    fl = f32(np.floor(x))
    alpha = f32(x - fl)
    i0 = int(fl)
    i1 = i0 + 1
    return min(max(i0, 0), size - 1), min(max(i1, 0), size - 1), alpha


def _linear_taps_repeat(coord, size: int):
    """§16.7 / wrapping REPEAT.  The coordinate is first reduced to [0, 1) (s - floor(s): REPEAT is periodic, this keeps
    fp32 precision for far-away uv), then i0 = floor(u - 0.5), i1 = i0 + 1, both taken mod size."""
    s = f32(coord)
    s = f32(s - f32(np.floor(s)))
    x = f32(s * f32(size) - f32(0.5))
    fl = f32(np.floor(x))
    alpha = f32(x - fl)
    if not np.isfinite(fl):
        return 0, 0, alpha
    i0 = int(fl)
    i1 = i0 + 1
    return i0 % size, i1 % size, alpha


def _bilinear(level: np.ndarray, x0, x1, y0, y1, ax, by) -> np.ndarray:
    t00, t10, t01, t11 = level[y0, x0], level[y0, x1], level[y1, x0], level[y1, x1]
    top = (t00 + ((t10 - t00).astype(np.float32) * ax).astype(np.float32)).astype(np.float32)
    bot = (t01 + ((t11 - t01).astype(np.float32) * ax).astype(np.float32)).astype(np.float32)
    return (top + ((bot - top).astype(np.float32) * by).astype(np.float32)).astype(np.float32)


def _levels_of(lod, count: int):
    """§16.5.9 with lod_min = 0, lod_max = LOD_CLAMP_NONE, MIPMAP_MODE_LINEAR: (d_hi, d_lo, delta)."""
    lam = f32(lod)
    lam = f32(min(max(lam, f32(0.0)), f32(count - 1))) if lam == lam else f32(0.0)
    fl = f32(np.floor(lam))
    d_hi = int(fl)
    return d_hi, min(d_hi + 1, count - 1), f32(lam - fl)


def sample_pyramid(levels: Sequence[np.ndarray], u, v, lod) -> np.ndarray:
    """OpImageSampleExplicitLod of the RGBA16F opaque pyramid through the reference's `clamp_sampler`
    (src/main.rs:694-705: LINEAR / LINEAR / MIPMAP_MODE_LINEAR, CLAMP_TO_EDGE).  `levels`: (h_l, w_l, 4) float16."""
    d_hi, d_lo, delta = _levels_of(lod, len(levels))
    taps = []
    for d in (d_hi, d_lo):
        lv = levels[d]
        x0, x1, ax = _linear_taps_clamp(u, lv.shape[1])
        y0, y1, by = _linear_taps_clamp(v, lv.shape[0])
        taps.append(_bilinear(lv.astype(np.float32) if lv.size <= 64 else _F32View(lv), x0, x1, y0, y1, ax, by))
    a, b = taps
    return (a + ((b - a).astype(np.float32) * delta).astype(np.float32)).astype(np.float32)


class _F32View:
    """Indexing view that converts just the fetched half texel to fp32 (no whole-level conversion per sample)."""

    def __init__(self, level):
        self.level = level

    def __getitem__(self, idx):
        return self.level[idx].astype(np.float32)


def sample_lut(rgba8: np.ndarray, u, v) -> np.ndarray:
    """OpImageSampleImplicitLod of the single-level R8G8B8A8_UNORM GGX LUT through `clamp_sampler`: bilinear."""
    h, w = rgba8.shape[:2]
    x0, x1, ax = _linear_taps_clamp(u, w)
    y0, y1, by = _linear_taps_clamp(v, h)

    class _D:
        def __getitem__(self, idx):
            return rgba8[idx].astype(np.float32) / f32(255.0)
    return _bilinear(_D(), x0, x1, y0, y1, ax, by)


def sample_texture(levels: Sequence[np.ndarray], srgb: bool, u, v, duv_dx, duv_dy) -> np.ndarray:
    """OpImageSampleImplicitLod of a material texture through the reference's `sampler` (src/main.rs:683-692: LINEAR /
    LINEAR / MIPMAP_MODE_LINEAR, address modes left at REPEAT): implicit LOD from the quad's uv derivatives."""
    w, h = levels[0].shape[1], levels[0].shape[0]
    mxx, mxy = f32(duv_dx[0]) * f32(w), f32(duv_dx[1]) * f32(h)
    myx, myy = f32(duv_dy[0]) * f32(w), f32(duv_dy[1]) * f32(h)
    rho_x = f32(np.sqrt(f32(f32(mxx * mxx) + f32(mxy * mxy))))
    rho_y = f32(np.sqrt(f32(f32(myx * myx) + f32(myy * myy))))
    rho = max(rho_x, rho_y)
    with np.errstate(divide="ignore", invalid="ignore"):
        lam = f32(_libm.log2f(float(rho)))
    d_hi, d_lo, delta = _levels_of(lam, len(levels))
    taps = []
    for d in (d_hi, d_lo):
        lv = levels[d]
        x0, x1, ax = _linear_taps_repeat(u, lv.shape[1])
        y0, y1, by = _linear_taps_repeat(v, lv.shape[0])

        class _D:
            def __getitem__(self, idx, lv=lv):
                return decode_rgba8(lv[idx], srgb)
        taps.append(_bilinear(_D(), x0, x1, y0, y1, ax, by))
    a, b = taps
    return (a + ((b - a).astype(np.float32) * delta).astype(np.float32)).astype(np.float32)
