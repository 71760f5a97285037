#!/usr/bin/env python3
"""Generates tests/golden/spirv_case_*.npz: outputs of the REFERENCE'S OWN compiled shaders
(/root/reference/compiled-shaders/normal/{fragment,fragment_transmission}.spv) executed by
oracle/spirv_ref/spirv_interp.py on seeded synthetic inputs.  Run in the authoring container only (the GPU box
has no /root/reference); the committed fixtures hold inputs and expected outputs, never shader text or binaries.

NO OUTPUT OF THE C ORACLE ENTERS A FIXTURE.  The fixed-function steps the SPIR-V delegates to the Vulkan implementation
(OpImageSample*: texel filtering, implicit LOD) and the mip chains the samplers read are answered by
oracle/spirv_ref/vk_sampling.py — a numpy fp32 statement of the Vulkan specification's equations, written from the
specification, with the choices the specification leaves open (operation order, weight precision, half rounding) listed
in its header — and OpDPdx / OpDPdy by the 2x2-quad differences below (Vulkan 1.3 "Derivative Operations", fine
derivatives of a fully covered quad).  So a fixture is "the reference's binary + the Vulkan specification";
tests/test_vk_sampling.py checks separately that the oracle's samplers are bit-identical to vk_sampling.
Transcendentals go through glibc (powf/logf/expf/log2f), the library the reference's CPU-side Rust would call.

cases a, b, c: every pixel of a 48x32 frame (2 lights; 4 lights + spotlights + ragged lists + roughness override;
               every texture slot with holes).
case f:        1 200 pixels of a 3840x2160 frame of TEXTURED materials (see sampled_4k_textured_case).
cases d, e:    2 000 random pixels of the BENCHMARK'S OWN frames at 3840x2160 — d: the headline scene (sun + 1 light),
               e: BASELINE config 3 (sun + 4 lights, roughness override 0.25) — whose framebuffer-size-dependent terms
               (lod = log2(3840) * r over a 12-level pyramid, 240x135-pixel clusters) the small frames never reach.  The
               fixture holds the sampled pixels' inputs and outputs; the GPU test shades the whole 4K frame and compares
               those pixels (tests/test_gpu_golden.py).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.spirv_ref import spirv_interp as si  # noqa: E402
from oracle.spirv_ref import vk_sampling as vk  # noqa: E402
from transmission_renderer_amd import synthetic, wire  # noqa: E402
from transmission_renderer_amd.png import read_png_rgba8  # noqa: E402

REF = "/root/reference/compiled-shaders/normal"
OUT = os.path.join(ROOT, "tests", "golden")

_libm = C.CDLL("libm.so.6")
for _n in ("powf", "logf", "expf", "log2f", "sinf", "cosf"):
    getattr(_libm, _n).restype = C.c_float
    getattr(_libm, _n).argtypes = [C.c_float] * (2 if _n == "powf" else 1)


class LibmInterp(si.Interp):
    """GLSL.std.450 transcendentals through glibc, like the C oracle."""

    def _ext(self, inst, x):
        f = {26: None, 28: _libm.logf, 27: _libm.expf, 30: _libm.log2f, 13: _libm.sinf, 14: _libm.cosf}.get(inst, False)
        if inst == 26:
            return si._map(lambda b, e: np.float32(_libm.powf(float(b), float(e))), x[0], x[1])
        if f:
            return si._map(lambda v: np.float32(f(float(v))), x[0])
        return super()._ext(inst, x)


def scene_buffers(scene):
    """The descriptor-set contents of src/descriptor_sets.rs as raw bytes."""
    mats = b"".join(bytes(m) for m in scene["materials"])
    lights = b"".join(bytes(l) for l in scene["lights"]) or bytes(48)
    return {
        (0, 2): mats,                                   # materials[]
        (0, 3): bytes(scene["uniforms"]),               # Uniforms
        (2, 0): lights,                                 # lights[]
        (2, 1): scene["cluster_counts"].tobytes(),      # cluster_light_counts[]
        (2, 2): scene["light_indices"].tobytes(),       # light_indices[]
    }


def quad_derivs(g, x, y, eye):
    """OpDPdx / OpDPdy as the oracle restates them on a G-buffer: differences inside the 2x2 quad, zero when the
    partner is outside the frame or not covered."""
    h, w = g["material_id"].shape
    nc = wire.NOT_COVERED
    d = {"dpos_dx": np.zeros(3, np.float32), "dpos_dy": np.zeros(3, np.float32),
         "duv_dx": np.zeros(2, np.float32), "duv_dy": np.zeros(2, np.float32)}
    xa, xb, ya, yb = x & ~1, x | 1, y & ~1, y | 1
    if xb < w and g["material_id"][y, xa] != nc and g["material_id"][y, xb] != nc:
        d["dpos_dx"] = (-(eye - g["pos_depth"][y, xb, :3])) - (-(eye - g["pos_depth"][y, xa, :3]))   # of -view_vector
        d["duv_dx"] = g["uv"][y, xb] - g["uv"][y, xa]
    if yb < h and g["material_id"][ya, x] != nc and g["material_id"][yb, x] != nc:
        d["dpos_dy"] = (-(eye - g["pos_depth"][yb, x, :3])) - (-(eye - g["pos_depth"][ya, x, :3]))
        d["duv_dy"] = g["uv"][yb, x] - g["uv"][ya, x]
    return d


def run_module(name, scene, g, levels, lut, pixels, textures=()):
    """One invocation of entry point `name` per pixel.  levels: the opaque pyramid (vk.blit_chain_rgba16f); textures:
    [(mip chain, srgb)] of the bindless array."""
    mod = si.Module(os.path.join(REF, name + ".spv"))
    lut_index = int(scene["uniforms"].ggx_lut_texture_index)
    cur = {}

    def sample(kind, image, sampler, coord, lod):
        if image[0] == (3, 0):      # the opaque pyramid (set 3 binding 0), sample_by_lod through clamp_sampler
            assert kind == "lod"
            v = vk.sample_pyramid(levels, coord[0], coord[1], lod)
            return [v[0], v[1], v[2], 1.0]
        assert image[0] == (0, 0) and kind == "implicit", (kind, image)
        if image[1] == lut_index and sampler[0] == (0, 4):   # clamp_sampler: the GGX LUT (single level)
            v = vk.sample_lut(lut, coord[0], coord[1])
            return [v[0], v[1], 0.0, 1.0]
        # a material texture through `sampler` (set 0 binding 1): implicit LOD from the quad's uv derivatives
        assert sampler[0] == (0, 1), sampler
        d = cur["d"]
        chain, srgb = textures[image[1]]
        return list(vk.sample_texture(chain, srgb, coord[0], coord[1], d["duv_dx"], d["duv_dy"]))

    def derivative(op, value):
        d = cur["d"]
        key = ("dpos_" if len(value) == 3 else "duv_") + op
        return [np.float32(x) for x in d[key]]

    bufs = scene_buffers(scene)
    push = bytes(scene["push"])
    outs = {}
    steps = 0
    for (y, x) in pixels:
        inputs = {0: g["pos_depth"][y, x, :3], 1: g["nrm_scale"][y, x, :3], 2: g["uv"][y, x],
                  3: int(g["material_id"][y, x]), 4: g["nrm_scale"][y, x, 3],
                  "FragCoord": [x + 0.5, y + 0.5, g["pos_depth"][y, x, 3], 1.0]}
        cur["d"] = quad_derivs(g, x, y, np.array(list(scene["push"].view_position), dtype=np.float32))
        it = LibmInterp(mod, name, bufs, push, inputs, sample, derivative)
        o = it.run()
        steps += it.steps
        for k, v in o.items():
            outs.setdefault(k, []).append(np.array(v, dtype=np.float32))
    return {k: np.stack(v) for k, v in outs.items()}, steps


def scene_arrays(scene):
    return dict(
        materials=np.frombuffer(b"".join(bytes(m) for m in scene["materials"]), dtype=np.uint8),
        lights=np.frombuffer(b"".join(bytes(l) for l in scene["lights"]), dtype=np.uint8),
        uniforms=np.frombuffer(bytes(scene["uniforms"]), dtype=np.uint8),
        push=np.frombuffer(bytes(scene["push"]), dtype=np.uint8),
        cluster_counts=scene["cluster_counts"],
        light_list=scene["light_indices"].reshape(-1, wire.MAX_LIGHTS_PER_CLUSTER)[0].copy())   # same in every cluster


def sampled_4k_case(tag, lut, num_point_lights, roughness_override, n_pixels=2000):
    """cases d / e: n random pixels of a benchmark frame at 3840x2160 (+ the frame's corners and a few pixels either side
    of material and cluster borders), through both entry points."""
    w, h = 3840, 2160
    scene = synthetic.make_scene(w, h, num_point_lights=num_point_lights, roughness_override=roughness_override)
    g = scene["gbuffer"]
    mip0 = synthetic.make_opaque_mip0(w, h)
    t0 = time.time()
    levels = vk.blit_chain_rgba16f(mip0)
    print(f"case {tag}: {len(levels)}-level pyramid of {w}x{h} in {time.time() - t0:.1f} s")
    rng = np.random.default_rng({"d": 0xD, "e": 0xE}[tag])
    ys = rng.integers(0, h, n_pixels - 40)
    xs = rng.integers(0, w, n_pixels - 40)
    pix = list(zip(ys.tolist(), xs.tolist()))
    pix += [(0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1)]
    mid = g["material_id"]
    bx = np.argwhere(mid[:, 1:] != mid[:, :-1])                      # material borders: both sides
    for k in rng.integers(0, len(bx), 9):
        y, x = bx[k]
        pix += [(int(y), int(x)), (int(y), int(x) + 1)]
    for k in range(9):                                               # cluster borders in x (240 px wide at 4K): both sides
        x = 240 * (k + 1)
        y = int(rng.integers(0, h))
        pix += [(y, x - 1), (y, x)]
    pix = pix[:n_pixels]
    t0 = time.time()
    out_t, steps = run_module("fragment_transmission", scene, g, levels, lut, pix)
    out_o, _ = run_module("fragment", scene, g, levels, lut, pix)
    print(f"case {tag}: {len(pix)} px of {w}x{h}, {steps / len(pix):.0f} SPIR-V instructions / px (transmission), "
          f"{time.time() - t0:.1f} s")
    py = np.array([p[0] for p in pix]), np.array([p[1] for p in pix])
    import hashlib
    np.savez_compressed(
        os.path.join(OUT, f"spirv_case_{tag}.npz"),
        width=w, height=h, num_point_lights=num_point_lights,
        roughness_override=np.float32(-1.0 if roughness_override is None else roughness_override),
        pixels=np.array(pix, dtype=np.int32),
        # the sampled pixels' inputs (the test regenerates the whole G-buffer with synthetic.make_gbuffer and checks these)
        pos_depth=g["pos_depth"][py], nrm_scale=g["nrm_scale"][py], uv=g["uv"][py], material_id=g["material_id"][py],
        opaque_mip0_sha256=np.frombuffer(hashlib.sha256(mip0.tobytes()).digest(), dtype=np.uint8),
        spirv_fragment_transmission=out_t[0], spirv_fragment_hdr=out_o[0], spirv_fragment_opaque_sampled=out_o[1],
        **scene_arrays(scene))


def sampled_4k_textured_case(tag, lut, n_pixels=1200):
    """case f: n pixels of a 3840x2160 frame whose materials bind texture slots (the lite class, the usual glTF set and the
    every-slot material of synthetic.apply_textures; sun + 2 lights), two thirds of them on textured materials: the
    full-class front end — implicit LOD from the 2x2 quad, normal mapping, per-lane roughness -> per-lane pyramid lod over
    the 12-level pyramid — at the frame size the small case c never reaches.  The fixture holds each sampled pixel's whole
    quad (the derivatives' inputs)."""
    w, h = 3840, 2160
    scene = synthetic.make_scene(w, h, num_point_lights=2, textured=True)
    scene["uniforms"].ggx_lut_texture_index = len(scene["textures"])
    g = scene["gbuffer"]
    textures = [(vk.blit_chain_rgba8(img, srgb), srgb) for img, srgb in scene["textures"]]
    mip0 = synthetic.make_opaque_mip0(w, h)
    t0 = time.time()
    levels = vk.blit_chain_rgba16f(mip0)
    print(f"case {tag}: {len(levels)}-level pyramid of {w}x{h} in {time.time() - t0:.1f} s")
    rng = np.random.default_rng(0xF)
    mid = g["material_id"]
    textured_ids = [i for i, m in enumerate(scene["materials"]) if any(getattr(m.textures, f) != -1 for f, _ in m.textures._fields_)]
    is_tex = np.isin(mid, np.array(textured_ids, dtype=mid.dtype))
    ty, tx = np.nonzero(is_tex)
    pick = rng.integers(0, len(ty), (2 * n_pixels) // 3)
    pix = list(zip(ty[pick].tolist(), tx[pick].tolist()))
    rest = n_pixels - len(pix)
    pix += list(zip(rng.integers(0, h, rest).tolist(), rng.integers(0, w, rest).tolist()))
    t0 = time.time()
    out_t, steps = run_module("fragment_transmission", scene, g, levels, lut, pix, textures)
    out_o, _ = run_module("fragment", scene, g, levels, lut, pix, textures)
    print(f"case {tag}: {len(pix)} px of {w}x{h} ({int(is_tex[[p[0] for p in pix], [p[1] for p in pix]].sum())} on textured materials), "
          f"{steps / len(pix):.0f} SPIR-V instructions / px (transmission), {time.time() - t0:.1f} s")
    py, px_ = np.array([p[0] for p in pix]), np.array([p[1] for p in pix])
    qy, qx = (py & ~1)[:, None, None] + np.arange(2)[None, :, None], (px_ & ~1)[:, None, None] + np.arange(2)[None, None, :]
    import hashlib
    np.savez_compressed(
        os.path.join(OUT, f"spirv_case_{tag}.npz"),
        width=w, height=h, num_point_lights=2, roughness_override=np.float32(-1.0), textured=1,
        pixels=np.array(pix, dtype=np.int32),
        pos_depth=g["pos_depth"][py, px_], nrm_scale=g["nrm_scale"][py, px_], uv=g["uv"][py, px_], material_id=g["material_id"][py, px_],
        quad_pos_depth=g["pos_depth"][qy, qx], quad_nrm_scale=g["nrm_scale"][qy, qx], quad_uv=g["uv"][qy, qx],
        quad_material_id=g["material_id"][qy, qx],
        opaque_mip0_sha256=np.frombuffer(hashlib.sha256(mip0.tobytes()).digest(), dtype=np.uint8),
        texture_srgb=np.array([s_ for _, s_ in scene["textures"]], dtype=np.uint8),
        **{f"texture_{i}": img for i, (img, _) in enumerate(scene["textures"])},
        spirv_fragment_transmission=out_t[0], spirv_fragment_hdr=out_o[0], spirv_fragment_opaque_sampled=out_o[1],
        **scene_arrays(scene))


def main():
    os.makedirs(OUT, exist_ok=True)
    lut = read_png_rgba8(os.path.join(ROOT, "transmission_renderer_amd", "assets", "ggx_lut.png"))
    w, h = 48, 32
    cases = {
        "a": dict(num_point_lights=2, roughness_override=None),
        "b": dict(num_point_lights=4, roughness_override=0.25),
        "c": dict(num_point_lights=2, roughness_override=None, textured=True, coverage="holes"),
    }
    only = sys.argv[1:]
    for tag, kw in cases.items():
        if only and tag not in only:
            continue
        scene = synthetic.make_scene(w, h, **kw)
        if tag == "c":   # textures repeat a few times across the frame; the GGX LUT is texture 8 of the bindless array
            scene["gbuffer"]["uv"] *= np.float32(0.75)
            scene["uniforms"].ggx_lut_texture_index = len(scene["textures"])
        textures = [(vk.blit_chain_rgba8(img, srgb), srgb) for img, srgb in scene.get("textures", [])]
        if tag == "b":   # the reference's spotlight rig in the opaque pass; per-cluster lists of different length
            scene["lights"] = wire.default_lights(spotlights=True)
            counts, idx = synthetic.all_lights_cluster_tables(4)
            rng = np.random.default_rng(7)
            counts[:] = rng.integers(0, 5, counts.size)
            scene["cluster_counts"], scene["light_indices"] = counts, idx
        g = scene["gbuffer"]
        levels = vk.blit_chain_rgba16f(synthetic.make_opaque_mip0(w, h))
        pixels = [(y, x) for y in range(h) for x in range(w)]
        t0 = time.time()
        if tag == "c":
            pixels = [(y, x) for (y, x) in pixels if g["material_id"][y, x] != wire.NOT_COVERED]
        out_t, steps = run_module("fragment_transmission", scene, g, levels, lut, pixels, textures)
        out_o, _ = run_module("fragment", scene, g, levels, lut, pixels, textures)
        print(f"case {tag}: {len(pixels)} px, {steps / len(pixels):.0f} SPIR-V instructions / px (transmission), "
              f"{time.time() - t0:.1f} s")
        np.savez_compressed(
            os.path.join(OUT, f"spirv_case_{tag}.npz"),
            width=w, height=h,
            pos_depth=g["pos_depth"], nrm_scale=g["nrm_scale"], uv=g["uv"], material_id=g["material_id"],
            opaque_mip0=synthetic.make_opaque_mip0(w, h),
            pixels=np.array(pixels, dtype=np.int32),
            texture_srgb=np.array([s_ for _, s_ in scene.get("textures", [])], dtype=np.uint8),
            **{f"texture_{i}": img for i, (img, _) in enumerate(scene.get("textures", []))},
            spirv_fragment_transmission=out_t[0],
            spirv_fragment_hdr=out_o[0],
            spirv_fragment_opaque_sampled=out_o[1],
            **scene_arrays(scene),
        )
    if not only or "d" in only:
        sampled_4k_case("d", lut, 1, None)
    if not only or "e" in only:
        sampled_4k_case("e", lut, 4, 0.25)
    if not only or "f" in only:
        sampled_4k_textured_case("f", lut)


if __name__ == "__main__":
    main()
