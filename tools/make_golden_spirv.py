#!/usr/bin/env python3
"""Generates tests/golden/spirv_*.npz: outputs of the REFERENCE'S OWN compiled shaders
(/root/reference/compiled-shaders/normal/{fragment,fragment_transmission}.spv) executed by
oracle/spirv_ref/spirv_interp.py on seeded synthetic inputs.  Run in the authoring container only (the GPU box
has no /root/reference); the committed fixtures hold inputs and expected outputs, never shader text or binaries.

Fixed-function steps that the SPIR-V delegates to the Vulkan implementation (OpImageSample*) are answered by the
oracle's restatement (o_sample_pyramid / o_sample_lut), so what the fixtures pin is everything else: every
arithmetic instruction, its order, the control flow (cluster light loop) and the buffer layouts.
Transcendentals go through the same libm the C oracle links (powf/logf/expf/log2f), so agreement is expected to
be bit-exact when the operation order is the same.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from oracle.spirv_ref import spirv_interp as si  # noqa: E402
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


def run_module(name, scene, g, tex, lut, pixels, binding=None):
    mod = si.Module(os.path.join(REF, name + ".spv"))
    L = oracle.load()
    w, h = g["width"], g["height"]
    pyr = oracle.pyramid_struct(w, h, tex)
    lut_p = lut.ctypes.data_as(C.c_void_p)
    lut_index = int(scene["uniforms"].ggx_lut_texture_index)
    cur = {}

    def sample(kind, image, sampler, coord, lod):
        if image[0] == (3, 0):      # the opaque pyramid (set 3 binding 0), sample_by_lod
            assert kind == "lod"
            v = L.o_sample_pyramid(C.byref(pyr), float(coord[0]), float(coord[1]), float(lod))
            return [v.x, v.y, v.z, 1.0]
        assert image[0] == (0, 0) and kind == "implicit", (kind, image)
        if image[1] == lut_index and sampler[0] == (0, 4):   # clamp_sampler: the GGX LUT (single level)
            v = L.o_sample_lut(lut_p, lut.shape[1], lut.shape[0], float(coord[0]), float(coord[1]))
            return [v.x, v.y, 0.0, 1.0]
        # a material texture through `sampler` (set 0 binding 1): implicit LOD from the quad's uv derivatives
        assert sampler[0] == (0, 1), sampler
        out = (C.c_float * 4)()
        d = cur["d"]
        L.o_sample_texture(C.byref(binding.texture_structs[image[1]]), float(coord[0]), float(coord[1]),
                           oracle.Vec2(float(d["duv_dx"][0]), float(d["duv_dx"][1])),
                           oracle.Vec2(float(d["duv_dy"][0]), float(d["duv_dy"][1])), C.byref(out))
        return list(out)

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
        binding = oracle.SceneBinding(scene, lut)
        if tag == "b":   # the reference's spotlight rig in the opaque pass; per-cluster lists of different length
            scene["lights"] = wire.default_lights(spotlights=True)
            counts, idx = synthetic.all_lights_cluster_tables(4)
            rng = np.random.default_rng(7)
            counts[:] = rng.integers(0, 5, counts.size)
            scene["cluster_counts"], scene["light_indices"] = counts, idx
        g = scene["gbuffer"]
        tex = oracle.new_pyramid(w, h, synthetic.make_opaque_mip0(w, h))
        oracle.generate_mips(w, h, tex)
        pixels = [(y, x) for y in range(h) for x in range(w)]
        t0 = time.time()
        if tag == "c":
            pixels = [(y, x) for (y, x) in pixels if g["material_id"][y, x] != wire.NOT_COVERED]
        out_t, steps = run_module("fragment_transmission", scene, g, tex, lut, pixels, binding)
        out_o, _ = run_module("fragment", scene, g, tex, lut, pixels, binding)
        print(f"case {tag}: {len(pixels)} px, {steps / len(pixels):.0f} SPIR-V instructions / px (transmission), "
              f"{time.time() - t0:.1f} s")
        np.savez_compressed(
            os.path.join(OUT, f"spirv_case_{tag}.npz"),
            width=w, height=h,
            materials=np.frombuffer(b"".join(bytes(m) for m in scene["materials"]), dtype=np.uint8),
            lights=np.frombuffer(b"".join(bytes(l) for l in scene["lights"]), dtype=np.uint8),
            uniforms=np.frombuffer(bytes(scene["uniforms"]), dtype=np.uint8),
            push=np.frombuffer(bytes(scene["push"]), dtype=np.uint8),
            cluster_counts=scene["cluster_counts"],
            light_list=scene["light_indices"].reshape(-1, wire.MAX_LIGHTS_PER_CLUSTER)[0].copy(),  # same in every cluster
            pos_depth=g["pos_depth"], nrm_scale=g["nrm_scale"], uv=g["uv"], material_id=g["material_id"],
            opaque_mip0=synthetic.make_opaque_mip0(w, h),
            pixels=np.array(pixels, dtype=np.int32),
            texture_srgb=np.array([s_ for _, s_ in scene.get("textures", [])], dtype=np.uint8),
            **{f"texture_{i}": img for i, (img, _) in enumerate(scene.get("textures", []))},
            spirv_fragment_transmission=out_t[0],
            spirv_fragment_hdr=out_o[0],
            spirv_fragment_opaque_sampled=out_o[1],
        )


if __name__ == "__main__":
    main()
