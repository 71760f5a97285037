"""Frustum culling + draw demultiplex: the oracle against the reference's compiled
frustum_culling.spv / demultiplex_draws.spv (tests/golden/spirv_culling.npz, made by tools/make_golden_culling.py),
plus host-side checks of the push constants."""
import ctypes as C
import os

import numpy as np

from oracle import oracle
from transmission_renderer_amd import meshes, wire

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spirv_culling.npz")


def test_record_layouts_match_the_shaders():
    # strides / offsets read from the OpMemberDecorate / ArrayStride of the compiled shaders
    assert wire.INSTANCE_DTYPE.itemsize == 48 and wire.INSTANCE_DTYPE.fields["primitive_id"][1] == 32
    assert wire.INSTANCE_DTYPE.fields["material_id"][1] == 36
    assert wire.PRIMITIVE_DTYPE.itemsize == 32 and wire.PRIMITIVE_DTYPE.fields["draw_buffer_index"][1] == 16
    assert wire.PRIMITIVE_DTYPE.fields["first_instance"][1] == 28
    assert wire.DRAW_COMMAND_DTYPE.itemsize == 20
    assert wire.CullingPushConstants.frustum_x_xz.offset == 64 and wire.CullingPushConstants.z_near.offset == 80


def test_culling_push_constants_host_side():
    """src/main.rs:1726-1746 in numpy (wire) and in the oracle agree; the planes are unit vectors."""
    z = np.load(GOLDEN)
    for k in range(2):
        a = wire.CullingPushConstants.new(z[f"perspective_{k}"], z[f"view_{k}"])
        b = oracle.culling_push_constants(z[f"perspective_{k}"], z[f"view_{k}"])
        assert bytes(a) == bytes(b) == z[f"push_{k}"].tobytes()
        for v in (a.frustum_x_xz, a.frustum_y_yz):
            assert abs(np.hypot(v[0], v[1]) - 1.0) < 1e-6


def test_frustum_culling_and_demultiplex_match_spirv():
    z = np.load(GOLDEN)
    prims, insts = z["primitives"], z["instances"]
    for k in range(2):
        push = wire.CullingPushConstants.from_buffer_copy(z[f"push_{k}"].tobytes())
        counts = oracle.frustum_culling(prims, insts, push)
        np.testing.assert_array_equal(counts, z[f"spirv_instance_counts_{k}"])
        dc, draws = oracle.demultiplex_draws(prims, counts)
        np.testing.assert_array_equal(dc, z[f"spirv_draw_counts_{k}"])
        for b in range(4):
            np.testing.assert_array_equal(draws[b], z[f"spirv_draws_{k}_{b}"].astype(wire.DRAW_COMMAND_DTYPE))
    # the fixture exercises both outcomes
    assert (z["spirv_instance_counts_0"] != z["spirv_instance_counts_1"]).any()
    assert (z["spirv_instance_counts_1"] == 0).any()


def test_culling_properties():
    """Size-independent properties: a sphere containing the camera is never culled, one far behind it always is,
    counts add up, and demultiplexing preserves (index_count, first_index, first_instance)."""
    rng = np.random.default_rng(3)
    scene = meshes.make_mesh_scene()
    prims = scene["primitives"].copy()
    n = 500
    insts = np.zeros(n, dtype=wire.INSTANCE_DTYPE)
    insts["translation_and_scale"][:, :3] = rng.uniform(-40, 40, (n, 3))
    insts["translation_and_scale"][:, 3] = rng.uniform(0.1, 3.0, n)
    q = rng.normal(size=(n, 4))
    insts["rotation"] = q / np.linalg.norm(q, axis=1, keepdims=True)
    insts["primitive_id"] = rng.integers(0, len(prims), n)
    eye, view = wire.default_camera()
    push = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(1920, 1080), view)
    counts = oracle.frustum_culling(prims, insts, push)
    assert counts.sum() <= n and counts.sum() > 0
    # per-instance decisions: cull each instance alone
    alone = np.array([oracle.frustum_culling(prims, insts[i:i + 1], push).sum() for i in range(n)])
    assert alone.sum() == counts.sum()
    np.testing.assert_array_equal(np.bincount(insts["primitive_id"][alone == 1], minlength=len(prims)), counts)
    behind = insts[:1].copy()
    behind["translation_and_scale"][0] = (0.0, 3.0, 50.0, 1.0)       # camera looks down -z from (0, 3, 1)
    assert oracle.frustum_culling(prims, behind, push).sum() == 0
    around = behind.copy()
    around["translation_and_scale"][0] = (eye[0], eye[1], eye[2], 100.0)
    assert oracle.frustum_culling(prims, around, push).sum() == 1
    dc, draws = oracle.demultiplex_draws(prims, counts)
    assert dc.sum() == (counts > 0).sum()
    for b in range(4):
        ids = np.nonzero((counts > 0) & (np.minimum(prims["draw_buffer_index"], 3) == b))[0]
        np.testing.assert_array_equal(draws[b]["first_index"], prims["first_index"][ids])
        np.testing.assert_array_equal(draws[b]["instance_count"], counts[ids])
        assert (draws[b]["vertex_offset"] == 0).all()
