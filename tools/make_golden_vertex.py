#!/usr/bin/env python3
"""tests/golden/spirv_vertex.npz: the reference's compiled vertex entry points (vertex_instanced,
vertex_instanced_with_scale, depth_pre_pass_instanced, depth_pre_pass_vertex_alpha_clip) and the alpha-clip
fragment shader (depth_pre_pass_alpha_clip) executed by oracle/spirv_ref on the procedural mesh scene.
Authoring container only (reads /root/reference); the fixture holds inputs + outputs."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
from oracle.spirv_ref import spirv_interp as si
from transmission_renderer_amd import meshes, synthetic, wire
from tools.make_golden_spirv import LibmInterp

REF = "/root/reference/compiled-shaders/normal"


def main():
    scene = meshes.make_mesh_scene()
    w, h = 640, 360
    push = wire.make_push_constants(w, h)
    pc = bytes(push)
    insts = scene["instances"]
    rng = np.random.default_rng(17)
    picks = [(int(i), int(v)) for i in rng.integers(0, len(insts), 40) for v in rng.integers(0, len(scene["position"]), 3)]
    out = {k: [] for k in ("position", "normal", "uv", "material_id", "scale", "clip", "clip_depth_only", "clip_alpha", "uv_alpha",
                           "material_alpha")}
    mods = {n: si.Module(os.path.join(REF, n + ".spv")) for n in
            ("vertex_instanced", "vertex_instanced_with_scale", "depth_pre_pass_instanced", "depth_pre_pass_vertex_alpha_clip")}
    bufs = {(1, 0): insts.tobytes()}
    for (i, v) in picks:
        inp = {0: scene["position"][v], 1: scene["normal"][v], 2: scene["uv"][v], "InstanceIndex": i}
        a = LibmInterp(mods["vertex_instanced"], "vertex_instanced", bufs, pc, inp).run()
        b = LibmInterp(mods["vertex_instanced_with_scale"], "vertex_instanced_with_scale", bufs, pc, inp).run()
        c = LibmInterp(mods["depth_pre_pass_instanced"], "depth_pre_pass_instanced", bufs, pc, {0: inp[0], "InstanceIndex": i}).run()
        d = LibmInterp(mods["depth_pre_pass_vertex_alpha_clip"], "depth_pre_pass_vertex_alpha_clip", bufs, pc,
                       {0: inp[0], 1: inp[2], "InstanceIndex": i}).run()   # locations are sequential: position 0, uv 1
        if not out["position"]:
            print("output keys:", list(a.keys()), list(b.keys()), list(c.keys()), list(d.keys()))
        # vertex_instanced and _with_scale must agree on the shared outputs
        for k in (0, 1, 2, 3):
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
        assert np.array_equal(np.asarray(a[("builtin", 0)]), np.asarray(b[("builtin", 0)]))
        out["position"].append(np.asarray(b[0], np.float32)); out["normal"].append(np.asarray(b[1], np.float32))
        out["uv"].append(np.asarray(b[2], np.float32)); out["material_id"].append(int(b[3])); out["scale"].append(np.float32(b[4]))
        out["clip"].append(np.asarray(b[("builtin", 0)], np.float32))
        out["clip_depth_only"].append(np.asarray(c[("builtin", 0)], np.float32))
        out["clip_alpha"].append(np.asarray(d[("builtin", 0)], np.float32))
        out["uv_alpha"].append(np.asarray(d[0], np.float32)); out["material_alpha"].append(int(d[1]))

    # ---- depth_pre_pass_alpha_clip: kill decisions on a textured material table
    mats = synthetic.apply_textures(synthetic.make_materials())
    mats[2].alpha_clipping_cutoff = 0.75
    mats[7].alpha_clipping_cutoff = 0.9
    mats[4].alpha_clipping_cutoff = 0.5            # untextured: diffuse_factor.w against the cutoff
    mats[4].diffuse_factor[3] = 0.4
    textures = synthetic.make_textures()
    sc = synthetic.make_scene(8, 8, num_point_lights=1)
    sc["materials"], sc["textures"] = mats, textures
    lut = np.zeros((4, 4, 4), np.uint8)
    binding = oracle.SceneBinding(sc, lut)
    L = oracle.load()
    mod = si.Module(os.path.join(REF, "depth_pre_pass_alpha_clip.spv"))
    mats_b = b"".join(bytes(m) for m in mats)
    cases, killed = [], []
    cur = {}

    def sample(kind, image, sampler, coord, lod):
        assert kind == "implicit" and image[0] == (0, 0) and sampler[0] == (0, 1)
        o = (C.c_float * 4)()
        L.o_sample_texture(C.byref(binding.texture_structs[image[1]]), float(coord[0]), float(coord[1]),
                           oracle.Vec2(*cur["ddx"]), oracle.Vec2(*cur["ddy"]), C.byref(o))
        return list(o)

    for n in range(300):
        m = int(rng.choice([2, 7, 4, 0]))
        uv = rng.uniform(-1, 3, 2).astype(np.float32)
        cur["ddx"] = [float(x) for x in rng.uniform(-0.02, 0.02, 2).astype(np.float32)]
        cur["ddy"] = [float(x) for x in rng.uniform(-0.02, 0.02, 2).astype(np.float32)]
        it = LibmInterp(mod, "depth_pre_pass_alpha_clip", {(0, 2): mats_b}, b"", {0: uv, 1: m}, sample, None)
        k = 1 if it.run().get("discard") else 0
        cases.append([m, uv[0], uv[1], *cur["ddx"], *cur["ddy"]])
        killed.append(k)
    print("alpha clip: killed", sum(killed), "of", len(killed))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "spirv_vertex.npz"),
                        instances=insts, push=np.frombuffer(pc, np.uint8), picks=np.array(picks, np.int32),
                        in_position=scene["position"][[v for _, v in picks]], in_normal=scene["normal"][[v for _, v in picks]],
                        in_uv=scene["uv"][[v for _, v in picks]],
                        **{"spirv_" + k: np.array(v) for k, v in out.items()},
                        alpha_materials=np.frombuffer(mats_b, np.uint8), alpha_cases=np.array(cases, np.float64),
                        spirv_alpha_killed=np.array(killed, np.uint8))


if __name__ == "__main__":
    main()
