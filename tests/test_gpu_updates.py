"""GPU tests of the per-frame sub-range updates and of the opaque pass that writes into the pyramid (run with -m gpu on an
MI355X; everything through the C ABI).

tr_update_lights / tr_update_instances: the reference rewrites two spotlights and one model's instances every frame through
mapped buffers (src/main.rs:1244-1261, 1316-1322); here the records travel inside kernel arguments.  The criterion is
bit-exactness against a context that was given the same records by a full upload.
tr_shade_opaque_pyramid: levels 0 and 1 from the opaque launch itself, the rest by tr_generate_mips_from — bit for bit
tr_shade_opaque + tr_generate_mips.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from transmission_renderer_amd import _lib, meshes, synthetic, wire  # noqa: E402
from test_gpu_raster import _scene  # noqa: E402


@pytest.fixture(scope="module")
def renderer(ggx_lut):
    if not torch.cuda.is_available():
        pytest.fail("no HIP device: the -m gpu tests must run on the GPU box")
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(ggx_lut)
    yield r
    r.close()


def _spot(pos, direction, outer, colour=(3.0, 2.5, 2.0)):
    d = np.asarray(direction, np.float32)
    return wire.Light.new_spot(pos, colour, 0.5, d / np.linalg.norm(d), outer * 0.8, outer)


def _frame(r, sc, geo, view, w, h, work=None):
    q = wire.view_rotation_inverse(view)
    culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
    work = work or r.new_frame_buffers(w, h)
    hdr, ldr = r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work)
    torch.cuda.synchronize()
    return hdr.clone(), ldr.clone(), work


def _rotated(instances, first, count, angle):
    """The reference's `instances[i].transform.rotation = Quat::from_rotation_y(angle)` (src/main.rs:1258-1261)."""
    out = instances.copy()
    out["rotation"][first:first + count] = np.array([0.0, np.sin(angle / 2), 0.0, np.cos(angle / 2)], np.float32)
    return out


def test_update_lights_and_instances_equal_full_uploads(renderer):
    r = renderer
    w, h = 640, 360
    view = wire.default_camera()[1]
    geo = meshes.make_mesh_scene()
    sc = _scene(w, h, view)
    lights = synthetic.make_lights(2) + [_spot((0.5, 2.5, -2.5), (0.0, -1.0, 0.2), 0.6), _spot((-1.0, 2.0, -3.5), (0.3, -1.0, 0.0), 0.5)]
    r.upload_materials(sc["materials"])
    r.upload_textures(sc["textures"])
    r.upload_lights(lights)
    r.upload_geometry(geo)
    base_hdr, _, work = _frame(r, sc, geo, view, w, h)

    # frame k of the reference's loop: both spotlights turned, one model rotated
    names = geo["instances"].dtype.names
    assert "rotation" in names and "primitive_id" in names
    first, count = 1, 2
    for k, angle in enumerate((0.35, 1.2, 2.9)):
        new_lights = list(lights)
        new_lights[2] = _spot((0.5, 2.5, -2.5), (np.sin(angle), -1.0, np.cos(angle)), 0.6)
        new_lights[3] = _spot((-1.0, 2.0, -3.5), (np.sin(angle + np.pi), -1.0, np.cos(angle + np.pi)), 0.5)
        new_instances = _rotated(geo["instances"], first, count, angle)
        r.update_lights(2, new_lights[2:4])
        r.update_instances(first, new_instances[first:first + count])
        got_hdr, got_ldr, _ = _frame(r, sc, geo, view, w, h, work)
        assert not torch.equal(got_hdr, base_hdr), "the updates changed nothing"
        # the same records by full uploads, another context
        from transmission_renderer_amd.renderer import TransmissionRenderer
        r2 = TransmissionRenderer(0)
        try:
            r2.upload_ggx_lut(None)
            r2.upload_materials(sc["materials"])
            r2.upload_textures(sc["textures"])
            r2.upload_lights(new_lights)
            r2.upload_geometry(dict(geo, instances=new_instances))
            want_hdr, want_ldr, _ = _frame(r2, sc, geo, view, w, h)
        finally:
            r2.close()
        assert torch.equal(got_hdr.view(torch.int16), want_hdr.view(torch.int16)), k
        assert torch.equal(got_ldr, want_ldr), k


def test_update_error_paths_and_large_ranges(renderer):
    r = renderer
    lights = synthetic.make_lights(70)     # more than one launch's payload (TR_UPDATE_MAX_BYTES / 96 = 32 lights)
    r.upload_lights(lights)
    lib = r.lib
    arr = wire.as_ctypes_array(lights, wire.Light)
    import ctypes as C
    assert lib.tr_update_lights(r._ctx, 0, 70, arr, None) == 0
    assert lib.tr_update_lights(r._ctx, 1, 70, arr, None) == 1          # past the uploaded array: TR_ERR_INVALID_ARGUMENT
    assert lib.tr_update_lights(r._ctx, 71, 0, arr, None) == 1
    assert lib.tr_update_lights(r._ctx, 3, 0, None, None) == 0          # an empty range is fine
    assert lib.tr_update_lights(r._ctx, 0, 1, None, None) == 1
    geo = meshes.make_mesh_scene()
    r.upload_geometry(geo)
    inst = np.ascontiguousarray(geo["instances"])
    n = len(inst)
    assert lib.tr_update_instances(r._ctx, 0, n, inst.ctypes.data, None) == 0
    assert lib.tr_update_instances(r._ctx, 1, n, inst.ctypes.data, None) == 1
    moved = inst.copy()
    other = [p for p in range(len(geo["primitives"])) if p != int(inst["primitive_id"][0])][0]
    moved["primitive_id"][0] = other
    assert lib.tr_update_instances(r._ctx, 0, 1, moved.ctypes.data, None) == 1   # an instance keeps its primitive
    torch.cuda.synchronize()


def test_updates_are_capturable(renderer):
    """The records are kernel arguments: an update can sit in a captured frame (a replay rewrites the baked records)."""
    r = renderer
    lights = synthetic.make_lights(4)
    r.upload_lights(lights)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            r.update_lights(1, [wire.Light.new_point((9.0, 8.0, 7.0), (1.0, 2.0, 3.0), 0.25)])
    r.update_lights(1, [lights[1]])
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()


@pytest.mark.parametrize("w,h", [(640, 360), (1920, 1080), (514, 290), (483, 273), (3840, 2160)])
def test_shade_opaque_pyramid_equals_shade_opaque_plus_generate_mips(renderer, w, h):
    from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid
    import bench
    r = renderer
    dev = r.device
    sc = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
    r.upload_materials(sc["materials"])
    r.upload_textures([])
    r.upload_lights(sc["lights"])
    r.set_cluster_tables(torch.from_numpy(sc["cluster_counts"].view(np.int32)).to(dev), torch.from_numpy(sc["light_indices"].view(np.int32)).to(dev))
    g = bench.make_gbuffer_torch(w, h, dev)
    want_p, got_p = OpaquePyramid(w, h, dev), OpaquePyramid(w, h, dev)
    want_hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=dev)
    got_hdr = torch.zeros_like(want_hdr)
    got_p.texels.fill_(float("nan"))
    r.shade_opaque(g, sc["uniforms"], sc["push"], want_hdr, want_p)
    r.generate_mips(want_p)
    nxt = r.shade_opaque_pyramid(g, sc["uniforms"], sc["push"], got_hdr, got_p)
    assert nxt == (2 if (w % 2 == 0 and h % 2 == 0) else 1)
    r.generate_mips_from(got_p, nxt)
    torch.cuda.synchronize()
    assert torch.equal(got_hdr.view(torch.int16), want_hdr.view(torch.int16))
    for l in range(want_p.levels):
        assert torch.equal(got_p.level(l).view(torch.int16), want_p.level(l).view(torch.int16)), l
    # a rect on odd pixels, or an RGBA32F target: level 0 only, the chain continues from level 1
    if w % 2 == 0 and h % 2 == 0:
        got_p.texels.fill_(float("nan"))
        gs = g.as_struct()
        import ctypes as C
        nxt = C.c_uint32()
        st = r.lib.tr_shade_opaque_pyramid(r._ctx, C.byref(gs), C.byref(sc["uniforms"]), C.byref(sc["push"]), got_hdr.data_ptr(),
                                           wire.FORMAT_RGBA16F, C.byref(got_p.desc), wire.Rect(0, 0, w - 1, h), C.byref(nxt), None)
        assert st == 0 and nxt.value == 1
        torch.cuda.synchronize()
