#!/usr/bin/env python3
"""tests/golden/spirv_culling.npz: the reference's compiled frustum_culling.spv and demultiplex_draws.spv executed by
oracle/spirv_ref on the procedural mesh scene (transmission_renderer_amd/meshes.py) from two cameras.  Authoring
container only (reads /root/reference); the fixture holds inputs + outputs.  Invocations run sequentially in
ascending order, so the atomically appended draw lists come out in ascending primitive order (on a GPU their
order is arbitrary; the set per draw buffer is what is pinned)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.spirv_ref import spirv_interp as si
from transmission_renderer_amd import meshes, wire
from tools.make_golden_spirv import LibmInterp

REF = "/root/reference/compiled-shaders/normal"


def run(scene, push):
    prims, insts = scene["primitives"], scene["instances"]
    counts = bytearray(4 * len(prims))
    mod = si.Module(os.path.join(REF, "frustum_culling.spv"))
    bufs = {(0, 0): prims.tobytes(), (0, 1): counts, (1, 0): insts.tobytes()}
    # one workgroup of 64 beyond the end as well: the `instance_id >= instances.len()` guard
    for i in range(((len(insts) + 63) // 64) * 64):
        LibmInterp(mod, "frustum_culling", bufs, bytes(push)[:84], {"GlobalInvocationId": [i, 0, 0]}).run()
    counts_a = np.frombuffer(bytes(counts), dtype=np.uint32).copy()
    mod = si.Module(os.path.join(REF, "demultiplex_draws.spv"))
    draw_counts = bytearray(16)
    draws = [bytearray(20 * len(prims)) for _ in range(4)]
    bufs = {(0, 0): prims.tobytes(), (0, 1): bytes(counts), (0, 2): draw_counts, (0, 3): draws[0], (0, 4): draws[1],
            (0, 5): draws[2], (0, 6): draws[3]}
    for d in range(((len(prims) + 63) // 64) * 64):
        LibmInterp(mod, "demultiplex_draws", bufs, b"", {"GlobalInvocationId": [d, 0, 0]}).run()
    dc = np.frombuffer(bytes(draw_counts), dtype=np.uint32).copy()
    return counts_a, dc, [np.frombuffer(bytes(b), dtype=wire.DRAW_COMMAND_DTYPE)[:dc[k]].copy() for k, b in enumerate(draws)]


def main():
    scene = meshes.make_mesh_scene()
    out = {"primitives": scene["primitives"], "instances": scene["instances"]}
    cams = [wire.default_camera()[1],
            wire.look_at_rh((3.5, 2.0, -6.0), (0.0, 1.2, -3.0), (0.0, 1.0, 0.0))]
    for k, view in enumerate(cams):
        proj = wire.perspective_matrix_reversed(1280, 720)
        push = wire.CullingPushConstants.new(proj, view)
        counts, dc, draws = run(scene, push)
        print("camera", k, "instance counts", counts, "draw counts", dc)
        out[f"push_{k}"] = np.frombuffer(bytes(push), dtype=np.uint8)
        out[f"perspective_{k}"] = proj
        out[f"view_{k}"] = view.astype(np.float32)
        out[f"spirv_instance_counts_{k}"] = counts
        out[f"spirv_draw_counts_{k}"] = dc
        for b in range(4):
            out[f"spirv_draws_{k}_{b}"] = draws[b]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "spirv_culling.npz"), **out)


if __name__ == "__main__":
    main()
