"""oracle/spirv_ref/vk_sampling.py — the numpy restatement of the Vulkan texel-filter / LOD / blit equations that
tools/make_golden_spirv.py uses to answer the reference shaders' OpImageSample* and to build the mip chains — against
the C oracle's samplers, BIT FOR BIT.  The fixtures are therefore "reference binary + Vulkan specification" without any
output of oracle/tr_oracle.c, and this test is what ties the oracle's fixed-function restatement to the same equations."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from oracle.spirv_ref import vk_sampling as vk
from transmission_renderer_amd import synthetic, wire


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("size", [(48, 32), (250, 130), (257, 131), (64, 64), (1, 1), (5, 3), (1920 // 4, 1080 // 4)])
def test_blit_chain_rgba16f_equals_oracle(size):
    w, h = size
    rng = np.random.default_rng(w * 1000 + h)
    mip0 = (rng.random((h, w, 4), dtype=np.float32) * 8.0 - 1.0).astype(np.float16)
    mip0[rng.integers(0, h), rng.integers(0, w)] = 60000.0          # a near-overflow texel
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    levels = vk.blit_chain_rgba16f(mip0)
    n, layout, total = wire.pyramid_layout(w, h)
    assert len(levels) == n == vk.mip_levels_for_size(w, h)
    for l, lv in enumerate(levels):
        off, lw, lh = layout[l]
        assert lv.shape == (lh, lw, 4)
        want = tex[off:off + lw * lh].reshape(lh, lw, 4)
        assert np.array_equal(lv.view(np.uint16), want.view(np.uint16)), f"level {l}"


def test_sample_pyramid_equals_oracle(ggx_lut):
    w, h = 250, 130
    mip0 = synthetic.make_opaque_mip0(w, h)
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    levels = vk.blit_chain_rgba16f(mip0)
    pyr = oracle.pyramid_struct(w, h, tex)
    L = oracle.load()
    rng = np.random.default_rng(5)
    coords = [(rng.uniform(-0.3, 1.3), rng.uniform(-0.3, 1.3), rng.uniform(-1.0, len(levels) + 1.0)) for _ in range(3000)]
    coords += [(0.0, 0.0, 0.0), (1.0, 1.0, 0.0), (0.5, 0.5, float(len(levels) - 1)), (1.0 - 1e-7, 1e-7, 2.5),
               (float("inf"), 0.5, 1.0), (0.5, float("-inf"), 1.0), (float("nan"), 0.25, 0.5), (0.25, 0.5, float("nan")),
               (123456.0, -98765.0, 3.25)]
    for u, v, lod in coords:
        u, v, lod = np.float32(u), np.float32(v), np.float32(lod)
        want = L.o_sample_pyramid(C.byref(pyr), float(u), float(v), float(lod))
        got = vk.sample_pyramid(levels, u, v, lod)
        assert np.array_equal(_bits(got[:3]), _bits([want.x, want.y, want.z])), (u, v, lod)


def test_sample_lut_equals_oracle(ggx_lut):
    L = oracle.load()
    lut_p = ggx_lut.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(6)
    coords = [(rng.uniform(-0.2, 1.2), rng.uniform(-0.2, 1.2)) for _ in range(3000)]
    coords += [(0.0, 0.0), (1.0, 1.0), (0.5 / 1024, 0.5 / 1024), (float("nan"), 0.5), (0.5, float("inf"))]
    for u, v in coords:
        u, v = np.float32(u), np.float32(v)
        want = L.o_sample_lut(lut_p, ggx_lut.shape[1], ggx_lut.shape[0], float(u), float(v))
        got = vk.sample_lut(ggx_lut, u, v)
        assert np.array_equal(_bits(got[:2]), _bits([want.x, want.y])), (u, v)


@pytest.mark.parametrize("srgb", [False, True])
def test_texture_chain_and_implicit_lod_sample_equal_oracle(srgb):
    rng = np.random.default_rng(11 + srgb)
    for (w, h) in ((64, 32), (37, 19), (8, 8)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        texels, t = oracle.make_texture(img, srgb)
        levels = vk.blit_chain_rgba8(img, srgb)
        off = 0
        for l, lv in enumerate(levels):
            lh, lw = lv.shape[:2]
            want = texels[off:off + lw * lh].reshape(lh, lw, 4)
            assert np.array_equal(lv, want), f"{w}x{h} srgb={srgb} level {l}"
            off += lw * lh
        assert off == texels.shape[0]
        L = oracle.load()
        for _ in range(600):
            u, v = np.float32(rng.uniform(-3.0, 3.0)), np.float32(rng.uniform(-3.0, 3.0))
            s = 10.0 ** rng.uniform(-4.0, 0.5)
            ddx = np.array([rng.normal() * s, rng.normal() * s], dtype=np.float32)
            ddy = np.array([rng.normal() * s, rng.normal() * s], dtype=np.float32)
            if rng.random() < 0.1:
                ddx[:] = 0.0
                ddy[:] = 0.0
            out = (C.c_float * 4)()
            L.o_sample_texture(C.byref(t), float(u), float(v), oracle.Vec2(float(ddx[0]), float(ddx[1])),
                               oracle.Vec2(float(ddy[0]), float(ddy[1])), C.byref(out))
            got = vk.sample_texture(levels, srgb, u, v, ddx, ddy)
            assert np.array_equal(_bits(got), _bits(list(out))), (w, h, srgb, u, v, ddx, ddy)
