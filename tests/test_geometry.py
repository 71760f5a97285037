"""Geometry front end, CPU side: the oracle's vertex stage and alpha-clip kill against the
reference's compiled shaders (tests/golden/spirv_vertex.npz, made by tools/make_golden_vertex.py), and the
restated rasteriser through properties (the fixed-function rules have no reference source: unpinned)."""
import os

import numpy as np
import pytest

from oracle import oracle
from transmission_renderer_amd import meshes, synthetic, wire

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spirv_vertex.npz")


def test_vertex_stage_matches_spirv_bit_exact():
    z = np.load(GOLDEN)
    push = wire.PushConstants.from_buffer_copy(z["push"].tobytes())
    pv = np.array(list(push.proj_view), dtype=np.float32)
    insts = z["instances"]
    for n, (i, _) in enumerate(z["picks"]):
        wp, wn, clip, scale = oracle.vertex_instanced(insts[i], pv, z["in_position"][n], z["in_normal"][n])
        np.testing.assert_array_equal(wp.astype(np.float32), z["spirv_position"][n])
        np.testing.assert_array_equal(wn.astype(np.float32), z["spirv_normal"][n])
        np.testing.assert_array_equal(clip, z["spirv_clip"][n])
        assert scale == z["spirv_scale"][n] and insts[i]["material_id"] == z["spirv_material_id"][n]
        # the depth-only and alpha-clip vertex shaders compute the same clip position; uv / material pass through
        np.testing.assert_array_equal(clip, z["spirv_clip_depth_only"][n])
        np.testing.assert_array_equal(clip, z["spirv_clip_alpha"][n])
        np.testing.assert_array_equal(z["in_uv"][n], z["spirv_uv"][n])
        np.testing.assert_array_equal(z["in_uv"][n], z["spirv_uv_alpha"][n])
        assert insts[i]["material_id"] == z["spirv_material_alpha"][n]


def test_alpha_clip_kill_matches_spirv():
    z = np.load(GOLDEN)
    n_mat = len(z["alpha_materials"]) // 160
    mats = [wire.MaterialInfo.from_buffer_copy(z["alpha_materials"][i * 160:(i + 1) * 160].tobytes()) for i in range(n_mat)]
    sc = synthetic.make_scene(8, 8, num_point_lights=1)
    sc["materials"], sc["textures"] = mats, synthetic.make_textures()
    b = oracle.SceneBinding(sc, np.zeros((4, 4, 4), np.uint8))
    got = [oracle.alpha_clip_kills(b, int(c[0]), c[1:3].astype(np.float32), c[3:5], c[5:7]) for c in z["alpha_cases"]]
    np.testing.assert_array_equal(np.array(got, np.uint8), z["spirv_alpha_killed"])
    assert 0 < z["spirv_alpha_killed"].sum() < len(got)


def _scene(w, h, view=None, textured=True):
    geo = meshes.make_mesh_scene()
    sc = synthetic.make_scene(w, h, num_point_lights=1, textured=textured, with_gbuffer=False)
    sc["materials"][2].alpha_clipping_cutoff = 0.75
    sc["materials"][7].alpha_clipping_cutoff = 0.6
    if view is not None:
        sc["push"] = wire.make_push_constants(w, h, view=view)
    return geo, sc


def _rasterize(geo, sc, w, h, view):
    b = oracle.SceneBinding(sc, np.zeros((4, 4, 4), np.uint8))
    push = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    counts = oracle.frustum_culling(geo["primitives"], geo["instances"], push)
    dc, draws = oracle.demultiplex_draws(geo["primitives"], counts)
    return oracle.rasterize(b, geo, dc, draws, w, h)


def test_rasterizer_properties():
    w, h = 160, 96
    _, view = wire.default_camera()
    geo, sc = _scene(w, h)
    opaque, trans = _rasterize(geo, sc, w, h, view)
    cov0 = opaque["material_id"] != wire.NOT_COVERED
    cov1 = trans["material_id"] != wire.NOT_COVERED
    assert 0.3 < cov0.mean() <= 1.0 and 0.02 < cov1.mean() < 0.6
    # depth in (0, 1], reversed-Z: a transmissive fragment only survives in front of the opaque surface
    d0, d1 = opaque["pos_depth"][..., 3], trans["pos_depth"][..., 3]
    assert (d0[cov0] > 0).all() and (d0[cov0] <= 1).all()
    assert (d1[cov1] > d0[cov1]).all()
    # materials: only those of the scene's instances, by layer
    prim_buf = geo["primitives"]["draw_buffer_index"][geo["instances"]["primitive_id"]]
    assert set(np.unique(opaque["material_id"][cov0])) <= set(geo["instances"]["material_id"][prim_buf < 2])
    assert set(np.unique(trans["material_id"][cov1])) <= set(geo["instances"]["material_id"][prim_buf >= 2])
    # interpolated world positions project back onto their own pixel, and frag_coord.z is their depth
    pv = np.array(list(sc["push"].proj_view), dtype=np.float64).reshape(4, 4).T      # [row][col]
    for layer, cov in ((opaque, cov0), (trans, cov1)):
        ys, xs = np.nonzero(cov)
        p = np.concatenate([layer["pos_depth"][ys, xs, :3].astype(np.float64), np.ones((len(ys), 1))], axis=1) @ pv.T
        sx, sy = (p[:, 0] / p[:, 3] * 0.5 + 0.5) * w, (p[:, 1] / p[:, 3] * 0.5 + 0.5) * h
        assert np.abs(sx - (xs + 0.5)).max() < 2e-2 and np.abs(sy - (ys + 0.5)).max() < 2e-2
        assert np.abs(p[:, 2] / p[:, 3] - layer["pos_depth"][ys, xs, 3]).max() < 1e-5
    # unit normals on spheres stay (about) unit after interpolation; model_scale is the instance's
    n = np.linalg.norm(opaque["nrm_scale"][..., :3][cov0], axis=1)
    assert (n > 0.9).all() and (n < 1.0 + 1e-5).all()
    assert set(np.unique(trans["nrm_scale"][..., 3][cov1])) <= set(geo["instances"]["translation_and_scale"][:, 3])
    # the alpha-clipped quads have holes (fragments killed by the diffuse texture's alpha)
    quad = opaque["material_id"] == 2
    assert 0 < quad.sum()
    sc2 = dict(sc)
    sc2["materials"] = [m for m in sc["materials"]]
    sc2["materials"][2].alpha_clipping_cutoff = 0.0
    opaque_nc, _ = _rasterize(geo, sc2, w, h, view)
    assert (opaque_nc["material_id"] == 2).sum() > quad.sum()


def test_rasterizer_watertight_and_order_independent():
    """Two triangles sharing an edge cover every pixel of their union exactly once (no cracks, no double hits:
    the tie rule), whatever the vertex order; back faces are culled."""
    w, h = 64, 48
    _, view = wire.default_camera()
    sc = synthetic.make_scene(w, h, num_point_lights=1, with_gbuffer=False)
    base = meshes.plane(3.0, 3.0, cells=3)
    S = meshes.Similarity
    tilt = meshes.quat_from_axis_angle([1, 0.2, 0.1], 1.1)
    results = []
    for perm in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        mesh = meshes.Mesh(base.position, base.normal, base.uv, base.index.reshape(-1, 3)[:, perm].reshape(-1))
        mb = meshes.ModelBuffers()
        mb.add_primitive(mesh, 0, [(S(np.array([0.1, 2.0, -2.5], np.float32), 1.0, tilt), 3)])
        geo = mb.finish()
        opaque, _ = _rasterize(geo, sc, w, h, view)
        results.append(opaque)
    for r in results[1:]:
        np.testing.assert_array_equal(r["material_id"], results[0]["material_id"])
        np.testing.assert_allclose(r["pos_depth"], results[0]["pos_depth"], rtol=0, atol=2e-5)
    cov = results[0]["material_id"] != wire.NOT_COVERED
    assert cov.sum() > 200
    # no cracks: the covered region of a convex quad grid is convex per row (one run of pixels)
    for y in range(h):
        xs = np.nonzero(cov[y])[0]
        if len(xs):
            assert len(xs) == xs[-1] - xs[0] + 1
    # flipped winding: everything is culled
    flipped = meshes.Mesh(base.position, base.normal, base.uv, base.index.reshape(-1, 3)[:, ::-1].reshape(-1))
    mb = meshes.ModelBuffers()
    mb.add_primitive(flipped, 0, [(S(np.array([0.1, 2.0, -2.5], np.float32), 1.0, tilt), 3)])
    opaque, _ = _rasterize(mb.finish(), sc, w, h, view)
    assert (opaque["material_id"] == wire.NOT_COVERED).all()


def test_rasterizer_near_plane_crossing():
    """A floor that runs through the camera plane (vertices with w <= 0): homogeneous edge functions need no
    clipper; fragments beyond the near plane (z > w) are dropped, the rest is shaded with finite attributes."""
    w, h = 96, 64
    eye, view = wire.default_camera()
    sc = synthetic.make_scene(w, h, num_point_lights=1, with_gbuffer=False)
    mb = meshes.ModelBuffers()
    mb.add_primitive(meshes.plane(40.0, 40.0, cells=1), 0, [(meshes.Similarity(np.array([0.0, 1.0, 0.0], np.float32)), 1)])
    opaque, _ = _rasterize(mb.finish(), sc, w, h, view)
    cov = opaque["material_id"] != wire.NOT_COVERED
    assert cov[h - 1].all() and not cov[0].any()          # floor below the horizon only
    assert np.isfinite(opaque["pos_depth"][cov]).all()
    assert np.abs(opaque["pos_depth"][..., 1][cov] - 1.0).max() < 1e-3     # every fragment lies on y = 1
