"""Clustered-light build: oracle pinned bit for bit against the reference's compiled
write_cluster_data.spv / assign_lights_to_clusters.spv (tests/golden/spirv_clusters.npz, generator
tools/make_golden_clusters.py); GPU kernels bit-identical to the oracle (-m gpu)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle
from transmission_renderer_amd import wire

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spirv_clusters.npz")


def _fixture():
    z = np.load(FIX)
    u = wire.Uniforms.from_buffer_copy(z["uniforms"].tobytes())
    n = z["lights"].size // 48
    lights = [wire.Light.from_buffer_copy(z["lights"].tobytes(), i * 48) for i in range(n)]
    return z, u, lights


def _lists_equal(counts, indices, want_counts, want_lists):
    np.testing.assert_array_equal(counts, want_counts)
    lists = indices.reshape(-1, wire.MAX_LIGHTS_PER_CLUSTER)
    for c in np.nonzero(want_counts)[0]:
        np.testing.assert_array_equal(lists[c, :want_counts[c]], want_lists[c, :want_counts[c]])


def test_oracle_matches_reference_spirv_clusters():
    z, u, lights = _fixture()
    aabbs = oracle.write_cluster_data(u, z["inverse_perspective"], (int(z["width"]), int(z["height"])))
    np.testing.assert_array_equal(aabbs.view(np.uint32), z["spirv_cluster_aabbs"].view(np.uint32))   # bit for bit
    counts, indices = oracle.assign_lights_to_clusters(lights, z["spirv_cluster_aabbs"], z["view_matrix"], z["view_rotation"])
    _lists_equal(counts, indices, z["spirv_counts"], z["spirv_lists"])
    # the fixture is not trivial: empty clusters, full clusters, spotlights culled from some clusters
    assert z["spirv_counts"].min() == 0 and z["spirv_counts"].max() >= 5
    spot = [i for i, l in enumerate(lights) if l.spotlight_direction_and_outer_angle[3] != 0]
    assert spot and any(((z["spirv_lists"] == s).any(axis=1) & (z["spirv_counts"] > 0)).sum() < wire.NUM_CLUSTERS for s in spot)


def test_cluster_aabbs_tile_the_frustum():
    u = wire.make_uniforms(1920, 1080)
    aabbs = oracle.write_cluster_data(u, wire.inverse_perspective(1920, 1080), (1920, 1080))
    a = aabbs.reshape(16, 16, 24, 8)
    assert (a[..., 0:3] <= a[..., 4:7]).all()
    # slices go away from the camera (view space looks down -z), x grows with the column index
    assert (np.diff(a[:, 8, 12, 2]) < 0).all() and (np.diff(a[5, 8, :, 0]) > 0).all()
    np.testing.assert_allclose(a[0, :, :, 6].max(), -0.01, rtol=1e-5)            # nearest slice starts at z_near
    np.testing.assert_allclose(a[15, :, :, 2].min(), -500.0, rtol=1e-4)          # farthest ends at z_far


def test_all_lights_table_is_what_big_falloff_produces():
    """synthetic.all_lights_cluster_tables == assign_lights_to_clusters when every falloff sphere covers the view."""
    from transmission_renderer_amd import synthetic
    _, view = wire.default_camera()
    u = wire.make_uniforms(640, 360)
    aabbs = oracle.write_cluster_data(u, wire.inverse_perspective(640, 360), (640, 360))
    lights = [wire.Light.new_point((0, 2, -2), (1, 1, 1), 1e9), wire.Light.new_point((1, 2, -3), (1, 0, 0), 1e9)]
    counts, indices = oracle.assign_lights_to_clusters(lights, aabbs, view, wire.view_rotation_inverse(view))
    c2, i2 = synthetic.all_lights_cluster_tables(2)
    np.testing.assert_array_equal(counts, c2)
    np.testing.assert_array_equal(indices, i2)


@pytest.mark.gpu
def test_gpu_cluster_build_bit_exact(ggx_lut):
    import torch
    from transmission_renderer_amd.renderer import TransmissionRenderer
    z, u, lights = _fixture()
    r = TransmissionRenderer(0)
    r.upload_lights(lights)
    aabbs = r.write_cluster_data(u, z["inverse_perspective"], (int(z["width"]), int(z["height"])))
    counts, indices = r.assign_lights_to_clusters(z["view_matrix"], z["view_rotation"], aabbs, bind=False)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(aabbs.cpu().numpy().view(np.uint32), z["spirv_cluster_aabbs"].view(np.uint32))
    _lists_equal(counts.cpu().numpy().view(np.uint32), indices.cpu().numpy().view(np.uint32), z["spirv_counts"], z["spirv_lists"])

    # 300 random lights (several 64-light chunks per wave, lists longer than one chunk), other resolution
    rng = np.random.default_rng(11)
    many = []
    for i in range(300):
        pos = (rng.uniform(-6, 6), rng.uniform(0, 5), rng.uniform(-12, 2))
        if i % 3 == 0:
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            many.append(wire.Light.new_spot(pos, rng.uniform(0, 1, 3), rng.uniform(0.2, 30), d, rng.uniform(0.1, 0.5), rng.uniform(0.5, 1.2)))
        else:
            many.append(wire.Light.new_point(pos, rng.uniform(0, 1, 3), rng.uniform(0.01, 3)))
    _, view = wire.default_camera()
    u2 = wire.make_uniforms(1000, 600)
    ip = wire.inverse_perspective(1000, 600)
    q = wire.view_rotation_inverse(view)
    want_aabbs = oracle.write_cluster_data(u2, ip, (1000, 600))
    want_counts, want_idx = oracle.assign_lights_to_clusters(many, want_aabbs, view, q)
    r.upload_lights(many)
    a2 = r.write_cluster_data(u2, ip, (1000, 600))
    c2, i2 = r.assign_lights_to_clusters(view, q, a2, bind=False)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a2.cpu().numpy().view(np.uint32), want_aabbs.view(np.uint32))
    assert want_counts.max() > 64
    _lists_equal(c2.cpu().numpy().view(np.uint32), i2.cpu().numpy().view(np.uint32), np.minimum(want_counts, 128),
                 want_idx.reshape(-1, 128))
    r.close()
