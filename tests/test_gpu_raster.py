"""GPU parity of the geometry front end (run with -m gpu on an MI355X): vertex stage,
rasterisation into the two TGB-v1 layers and the alpha-clip kill through the C ABI against the CPU oracle, whose
vertex stage / kill logic are pinned bit-exactly by the reference's compiled shaders (tests/test_geometry.py).

Criteria (see _compare): coverage, material ids and depth are BIT-EXACT (index work: same fp32 operations in the same
order as the oracle's rasteriser); the interpolated position / normal / uv agree within ATTRIBUTE_TOLERANCE (the
kernels evaluate the perspective interpolation as per-triangle plane equations, not as the oracle's barycentric mix).
The only other tolerance is on alpha-clipped primitives, where the kill compares a filtered texture value (computed
with v_log_f32 / fma on the GPU) against the cutoff: a handful of pixels on the cut-out's rim may differ.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from transmission_renderer_amd import meshes, synthetic, wire  # noqa: E402


@pytest.fixture(scope="module")
def renderer(ggx_lut):
    if not torch.cuda.is_available():
        pytest.fail("no HIP device: the -m gpu tests must run on the GPU box")
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(ggx_lut)
    yield r
    r.close()


def _scene(w, h, view, alpha_cutoffs=(0.75, 0.6)):
    sc = synthetic.make_scene(w, h, num_point_lights=1, textured=True, with_gbuffer=False)
    sc["materials"][2].alpha_clipping_cutoff = alpha_cutoffs[0]
    sc["materials"][7].alpha_clipping_cutoff = alpha_cutoffs[1]
    eye = np.linalg.inv(np.asarray(view, np.float64).T)[:3, 3]
    sc["push"] = wire.make_push_constants(w, h, eye=eye.astype(np.float32), view=view)
    return sc


def _oracle_layers(geo, sc, w, h, view, fp64=False):
    b = oracle.SceneBinding(sc, np.zeros((4, 4, 4), np.uint8))
    push = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    counts = oracle.frustum_culling(geo["primitives"], geo["instances"], push)
    dc, draws = oracle.demultiplex_draws(geo["primitives"], counts)
    return oracle.rasterize(b, geo, dc, draws, w, h, fp64=fp64), push


def _gpu_layers(r, geo, sc, w, h, culling):
    r.upload_materials(sc["materials"])
    r.upload_textures(sc["textures"])
    r.upload_geometry(geo)
    o, t = r.new_layer(w, h), r.new_layer(w, h)
    r.draw_scene(culling, sc["push"], o, t)
    torch.cuda.synchronize()
    out = []
    for layer in (o, t):
        out.append({"pos_depth": layer.pos_depth.cpu().numpy(), "nrm_scale": layer.nrm_scale.cpu().numpy(),
                    "uv": layer.uv.cpu().numpy(), "material_id": layer.material_id.cpu().numpy().view(np.uint32)})
    return out


ATTRIBUTE_RMSE = 1e-4    # RMSE of (gpu - oracle) / max(|oracle|, 1) per plane over the pixels both sides cover: the criterion
ATTRIBUTE_MAX = 2e-3     # ... and its largest single value (the fp32 oracle's own rounding noise at sliver triangles, below)
_worst = {"rmse": 0.0, "max": 0.0}


def _attribute_errors(a, b, mask):
    """(rmse, max) of (a - b) / max(|b|, 1) over position / normal / uv of the masked pixels."""
    worst_rmse = worst_max = 0.0
    for k, n in (("pos_depth", 3), ("nrm_scale", 3), ("uv", 2)):
        x, y = a[k][..., :n].astype(np.float64)[mask], b[k][..., :n].astype(np.float64)[mask]
        if x.size == 0:
            continue
        assert np.isfinite(x).all(), k
        err = np.abs(x - y) / np.maximum(np.abs(y), 1.0)
        worst_rmse = max(worst_rmse, float(np.sqrt(np.mean(err ** 2))))
        worst_max = max(worst_max, float(err.max()))
    return worst_rmse, worst_max


def _compare(got, want, alpha_materials=(), max_rim_pixels=0, want64=None):
    """Two criteria.
    BIT-EXACT: coverage, material ids, depth (the rasteriser's words: index work) and the instance scale — except that
    up to `max_rim_pixels` pixels may differ where either side shows an alpha-clipped material (the pixel then shows
    whatever is behind the cut-out on the other side).
    WITHIN TOLERANCE: the interpolated position / normal / uv of every pixel both sides agree on — RMSE <= ATTRIBUTE_RMSE,
    no pixel beyond ATTRIBUTE_MAX.  The oracle mixes the corners with barycentrics (three fp32 edge functions evaluated
    about the frame origin, one IEEE division); the kernels evaluate the same rational function as plane equations
    centred on the triangle (tr_visibility.h: tr_tri_planes) with v_rcp_f32 — same function, different roundings: a few
    1e-7 typically; where the oracle's own edge functions cancel (sliver triangles far from the frame origin) ITS value
    moves by up to ~1e-4.  `want64` (the oracle's fp64 twin on the same scene) shows whose noise that is: the kernels
    must be as close to the fp64 evaluation as the fp32 oracle is."""
    differ = got["material_id"] != want["material_id"]
    covered = want["material_id"] != wire.NOT_COVERED        # without a fragment only the id plane is defined
    differ |= (got["pos_depth"][..., 3].view(np.uint32) != want["pos_depth"][..., 3].view(np.uint32)) & covered
    differ |= (got["nrm_scale"][..., 3].view(np.uint32) != want["nrm_scale"][..., 3].view(np.uint32)) & covered
    same = covered & ~differ
    rmse, worst = _attribute_errors(got, want, same)
    _worst["rmse"], _worst["max"] = max(_worst["rmse"], rmse), max(_worst["max"], worst)
    assert rmse <= ATTRIBUTE_RMSE and worst <= ATTRIBUTE_MAX, (rmse, worst)
    if want64 is not None:
        all3 = same & (want64["material_id"] == want["material_id"])
        g_rmse, g_max = _attribute_errors(got, want64, all3)
        o_rmse, o_max = _attribute_errors(want, want64, all3)
        print(f"   vs the fp64 twin: kernels rmse {g_rmse:.2e} max {g_max:.2e}; fp32 oracle rmse {o_rmse:.2e} max {o_max:.2e} "
              f"({int(all3.sum())} pixels); kernels vs fp32 oracle rmse {rmse:.2e} max {worst:.2e}")
        assert g_rmse <= 1.25 * o_rmse + 1e-7 and g_max <= 1.25 * o_max + 1e-6, (g_rmse, o_rmse, g_max, o_max)
    if not differ.any():
        return 0
    ys, xs = np.nonzero(differ)
    involved = np.isin(got["material_id"][ys, xs], alpha_materials) | np.isin(want["material_id"][ys, xs], alpha_materials)
    assert involved.all(), ("coverage / id / depth differ away from alpha-clipped surfaces", int((~involved).sum()), ys[~involved][:5], xs[~involved][:5])
    assert differ.sum() <= max_rim_pixels, int(differ.sum())
    return int(differ.sum())


CAMERAS = [
    ("default", lambda: wire.default_camera()[1]),
    ("side", lambda: wire.look_at_rh((3.5, 2.0, -6.0), (0.0, 1.2, -3.0), (0.0, 1.0, 0.0))),
    ("inside", lambda: wire.look_at_rh((0.1, 1.0, -2.6), (0.0, 1.4, -4.0), (0.0, 1.0, 0.0))),   # geometry crosses the near plane
]


@pytest.mark.parametrize("w,h", [(256, 256), (250, 130), (640, 360)])
@pytest.mark.parametrize("cam", [c[0] for c in CAMERAS])
def test_layers_coverage_depth_ids_bit_exact_attributes_within_tolerance(renderer, w, h, cam):
    view = dict(CAMERAS)[cam]()
    geo = meshes.make_mesh_scene()
    sc = _scene(w, h, view)
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    got_o, got_t = _gpu_layers(renderer, geo, sc, w, h, culling)
    budget = max(4, w * h // 20000)
    n0 = _compare(got_o, want_o, alpha_materials=(2,), max_rim_pixels=budget)
    n1 = _compare(got_t, want_t, alpha_materials=(7, 2), max_rim_pixels=budget)
    cov = want_o["material_id"] != wire.NOT_COVERED
    assert cov.mean() > 0.2, cov.mean()
    print(f"{cam} {w}x{h}: coverage {cov.mean():.2f} / {(want_t['material_id'] != wire.NOT_COVERED).mean():.2f}, "
          f"alpha-rim pixels {n0} + {n1}")


def test_layers_1080p(renderer):
    """BASELINE config 2's frame size: both rasterised layers against the oracle's — coverage, ids and depth bit for bit
    (alpha-clip rims counted as above), attributes within ATTRIBUTE_TOLERANCE."""
    w, h = 1920, 1080
    view = wire.default_camera()[1]
    geo = meshes.make_mesh_scene()
    sc = _scene(w, h, view)
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    got_o, got_t = _gpu_layers(renderer, geo, sc, w, h, culling)
    (twin_o, twin_t), _ = _oracle_layers(geo, sc, w, h, view, fp64=True)
    budget = w * h // 20000
    n0 = _compare(got_o, want_o, alpha_materials=(2,), max_rim_pixels=budget, want64=twin_o)
    n1 = _compare(got_t, want_t, alpha_materials=(7, 2), max_rim_pixels=budget, want64=twin_t)
    cov = want_o["material_id"] != wire.NOT_COVERED
    assert cov.mean() > 0.2, cov.mean()
    print(f"1080p: coverage {cov.mean():.2f} / {(want_t['material_id'] != wire.NOT_COVERED).mean():.2f}, alpha-rim pixels {n0} + {n1}, "
          f"attribute errors so far: rmse {_worst['rmse']:.2e} (bound {ATTRIBUTE_RMSE}), max {_worst['max']:.2e} (bound {ATTRIBUTE_MAX})")


def test_layers_without_alpha_clip_have_no_rim_tolerance(renderer):
    """Cutoff 0 switches every kill off: then there is no tolerance at all."""
    w, h = 512, 288
    view = wire.default_camera()[1]
    geo = meshes.make_mesh_scene()
    sc = _scene(w, h, view, alpha_cutoffs=(0.0, 0.0))
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    got_o, got_t = _gpu_layers(renderer, geo, sc, w, h, culling)
    assert _compare(got_o, want_o) == 0 and _compare(got_t, want_t) == 0


def test_many_small_triangles_and_depth_ties(renderer):
    """Finely tessellated spheres (sub-pixel triangles), coplanar duplicates (equal depths: the later-drawn triangle
    must win on both sides) and a large floor in one frame."""
    w, h = 384, 216
    view = wire.default_camera()[1]
    mb = meshes.ModelBuffers()
    S = meshes.Similarity
    f32 = np.float32
    mb.add_primitive(meshes.plane(30.0, 30.0, cells=2), 0, [(S(np.array([0, 0.6, -3.0], f32)), 3)])
    fine = meshes.uv_sphere(1.0, 160, 80)
    mb.add_primitive(fine, 0, [(S(np.array([-0.9, 1.6, -2.6], f32), 0.55), 1), (S(np.array([1.1, 1.4, -3.4], f32), 0.7), 6)])
    quad = meshes.plane(1.5, 1.5, cells=3)
    tilt = meshes.quat_from_axis_angle([1, 0, 0], 1.3)
    for mat in (8, 9, 10):            # three coplanar copies: material 10 is drawn last
        mb.add_primitive(quad, 0, [(S(np.array([0.0, 2.3, -3.0], f32), 1.0, tilt), mat)])
    mb.add_primitive(meshes.uv_sphere(1.0, 96, 48), 2, [(S(np.array([0.15, 1.7, -1.9], f32), 0.6), 4)])
    geo = mb.finish()
    sc = _scene(w, h, view, alpha_cutoffs=(0.0, 0.0))
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    (twin_o, twin_t), _ = _oracle_layers(geo, sc, w, h, view, fp64=True)
    got_o, got_t = _gpu_layers(renderer, geo, sc, w, h, culling)
    assert _compare(got_o, want_o, want64=twin_o) == 0 and _compare(got_t, want_t, want64=twin_t) == 0
    assert (want_o["material_id"] == 10).sum() > 50 and not (want_o["material_id"] == 8).any()


def test_long_thin_triangles_at_a_grazing_angle(renderer):
    """A floor of long slivers seen almost edge-on and coarse spheres close to the camera: most work items of such
    triangles' bounds hold no fragment (the rasteriser drops them in the lanes' preparation phase, tr_raster_kernels.h);
    coverage, depth and ids stay bit-exact, wide frame so that items span several 64-pixel tile columns."""
    w, h = 1024, 160
    _, view = wire.default_camera()
    mb = meshes.ModelBuffers()
    S = meshes.Similarity
    f32 = np.float32
    mb.add_primitive(meshes.plane(60.0, 60.0, cells=24), 0, [(S(np.array([0, 0.75, -8.0], f32)), 3)])
    coarse = meshes.uv_sphere(1.0, 10, 5)
    rng = np.random.default_rng(5)
    mb.add_primitive(coarse, 0, [(S(np.array([rng.uniform(-4, 4), rng.uniform(0.9, 2.5), rng.uniform(-9, -2)], f32), rng.uniform(0.3, 0.9)),
                                  int(rng.integers(0, 16))) for _ in range(12)])
    mb.add_primitive(coarse, 2, [(S(np.array([0.3, 1.5, -1.6], f32), 0.5), 4)])
    geo = mb.finish()
    sc = _scene(w, h, view, alpha_cutoffs=(0.0, 0.0))
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    (twin_o, twin_t), _ = _oracle_layers(geo, sc, w, h, view, fp64=True)
    got_o, got_t = _gpu_layers(renderer, geo, sc, w, h, culling)
    assert _compare(got_o, want_o, want64=twin_o) == 0 and _compare(got_t, want_t, want64=twin_t) == 0
    assert (want_o["material_id"] == 3).mean() > 0.1 and (want_t["material_id"] == 4).any()


def test_screen_filling_degenerate_and_off_screen_triangles(renderer):
    """Two triangles that fill the frame from close up (bounds clamped to the frame: the most work items a triangle can
    have), zero-area triangles (a strip of zero width: no work item may produce a fragment), a sphere whose instance passes
    the frustum test while most of its triangles lie off screen (empty bounds: zero work items in the prefix array) and a
    layer without any triangle: bit-exact."""
    w, h = 640, 360
    _, view = wire.default_camera()
    mb = meshes.ModelBuffers()
    S = meshes.Similarity
    f32 = np.float32
    wall = meshes.quat_from_axis_angle([1, 0, 0], np.pi / 2)       # the plane turned to face the camera
    mb.add_primitive(meshes.plane(40.0, 40.0, cells=1), 0, [(S(np.array([0, 1.5, -9.0], f32), 1.0, wall), 5)])
    mb.add_primitive(meshes.plane(0.0, 3.0, cells=4), 0, [(S(np.array([0, 1.5, -3.0], f32), 1.0, wall), 6)])
    mb.add_primitive(meshes.uv_sphere(1.0, 24, 12), 0, [(S(np.array([-6.4, 1.5, -5.0], f32), 1.6), 9), (S(np.array([0.4, 4.9, -4.0], f32), 1.2), 1)])
    geo = mb.finish()
    sc = _scene(w, h, view, alpha_cutoffs=(0.0, 0.0))
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    got_o, got_t = _gpu_layers(renderer, geo, sc, w, h, culling)
    assert _compare(got_o, want_o) == 0 and _compare(got_t, want_t) == 0
    ids = want_o["material_id"].view(np.int32)
    assert (ids != -1).all() and (ids == 5).mean() > 0.5 and not (ids == 6).any()     # the wall fills the frame, the strip draws nothing
    assert (ids == 9).any() and (ids == 1).any()                                       # both spheres reach into the frame
    assert (want_t["material_id"].view(np.int32) == -1).all()


def test_draw_scene_replays_from_a_hip_graph(renderer):
    """tr_draw_scene (culling, demultiplex, set-up with its look-back scan, rasteriser, resolve) captured into a HIP graph after
    a first call outside the capture: every replay writes the layers of the direct call (the scan's frame counter is on the
    device and moves on with every replay), and the context works on afterwards."""
    w, h = 256, 128
    view = wire.default_camera()[1]
    geo = meshes.make_mesh_scene()
    sc = _scene(w, h, view, alpha_cutoffs=(0.0, 0.0))
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    got_o, _ = _gpu_layers(renderer, geo, sc, w, h, culling)          # (uploads; a first, uncaptured call)
    o, t = renderer.new_layer(w, h), renderer.new_layer(w, h)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        renderer.draw_scene(culling, sc["push"], o, t)                  # (once on the capturing stream, outside the capture)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        renderer.draw_scene(culling, sc["push"], o, t)
    for k in range(3):
        o.material_id.fill_(-7)
        o.pos_depth.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(o.material_id.cpu().numpy().view(np.uint32), got_o["material_id"]), k
        covered = got_o["material_id"] != wire.NOT_COVERED              # (only the id plane is defined where no fragment landed)
        assert covered.mean() > 0.2 and np.array_equal(o.pos_depth.cpu().numpy()[covered], got_o["pos_depth"][covered]), k
    renderer.draw_scene(culling, sc["push"], o, t)
    torch.cuda.synchronize()
    assert np.array_equal(o.material_id.cpu().numpy().view(np.uint32), got_o["material_id"])
    assert _compare(got_o, want_o) == 0


def test_rasterize_then_shade_end_to_end(renderer, ggx_lut):
    """glTF-shaped path: geometry -> layers -> opaque pass -> mips -> transmissive pass, GPU vs the oracle doing the
    same from its own layers (T1-style bound on the final RGBA16F frame)."""
    from transmission_renderer_amd.renderer import OpaquePyramid
    r = renderer
    w, h = 320, 180
    view = wire.default_camera()[1]
    geo = meshes.make_mesh_scene()
    sc = _scene(w, h, view, alpha_cutoffs=(0.0, 0.0))
    sc["lights"] = synthetic.make_lights(2)
    sc["cluster_counts"], sc["light_indices"] = synthetic.all_lights_cluster_tables(2)
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    r.upload_materials(sc["materials"])
    r.upload_textures(sc["textures"])
    r.upload_lights(sc["lights"])
    r.set_cluster_tables(torch.from_numpy(sc["cluster_counts"].view(np.int32)).to(r.device),
                         torch.from_numpy(sc["light_indices"].view(np.int32)).to(r.device))
    r.upload_geometry(geo)
    o, t = r.new_layer(w, h), r.new_layer(w, h)
    r.draw_scene(culling, sc["push"], o, t)
    pyr = OpaquePyramid(w, h, r.device)
    hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
    r.record(o, t, sc["uniforms"], sc["push"], hdr, pyr)
    torch.cuda.synchronize()
    b = oracle.SceneBinding(sc, ggx_lut)
    for layer in (want_o, want_t):
        layer["width"], layer["height"] = w, h
    o16, _, mip0 = oracle.shade_opaque(b, want_o, nthreads=8, fp64=True)
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    oracle.shade_transmission(b, want_t, tex, hdr_f16=o16, nthreads=8, fp64=True)
    got, want = hdr.cpu().numpy().astype(np.float64), o16.astype(np.float64)
    fin = np.isfinite(got).all(axis=2) & np.isfinite(want).all(axis=2)
    assert fin.mean() > 0.995
    e = np.where(fin[..., None], (got - want) / np.maximum(np.abs(want), 1.0), 0.0)
    rmse = np.sqrt((e[..., :3] ** 2).mean(axis=(0, 1))).max()
    # the layers are bit-identical on both sides, so this is the shading passes' bound (measured 3e-5: the opaque
    # pass flips ~25 texels of mip 0 by one half-precision step, everything downstream follows the oracle)
    assert rmse <= 1e-4, rmse
    assert np.quantile(np.abs(e), 0.999) <= 2e-3
    # ... and the same chain through the PINNED fp32 oracle (its own mip 0 and pyramid), the ill-conditioned pixels counted
    from test_gpu_parity import _p1_against_pinned
    p16, _, pmip0 = oracle.shade_opaque(b, want_o, nthreads=8)
    ptex = oracle.new_pyramid(w, h, pmip0)
    oracle.generate_mips(w, h, ptex)
    oracle.shade_transmission(b, want_t, ptex, hdr_f16=p16, nthreads=8)
    pin = p16.astype(np.float64)
    ok = fin & np.isfinite(pin).all(axis=2)
    _p1_against_pinned(got, pin, want, "geometry -> frame (RGBA16F) vs oracle32", covered=ok)


def test_raster_error_paths(ggx_lut):
    from transmission_renderer_amd import _lib
    from transmission_renderer_amd.renderer import TransmissionRenderer
    fresh = TransmissionRenderer(0)
    w, h = 64, 64
    view = wire.default_camera()[1]
    sc = _scene(w, h, view)
    geo = meshes.make_mesh_scene()
    culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    o, t = fresh.new_layer(w, h), fresh.new_layer(w, h)
    with pytest.raises(_lib.TrError) as e:      # no geometry yet
        fresh.draw_scene(culling, sc["push"], o, t)
    assert e.value.status == 4
    bad = dict(geo)
    bad["index"] = geo["index"].copy()
    bad["index"][5] = len(geo["position"]) + 3
    with pytest.raises(_lib.TrError) as e:      # an index beyond the vertex buffer is refused at upload
        fresh.upload_geometry(bad)
    assert e.value.status == 1
    fresh.upload_geometry(geo)
    with pytest.raises(_lib.TrError) as e:      # materials missing
        fresh.draw_scene(culling, sc["push"], o, t)
    assert e.value.status == 4
    fresh.upload_materials(sc["materials"])
    with pytest.raises(_lib.TrError) as e:      # textured materials without their textures
        fresh.draw_scene(culling, sc["push"], o, t)
    assert e.value.status == 1
    fresh.upload_textures(sc["textures"])
    fresh.draw_scene(culling, sc["push"], o, t)
    torch.cuda.synchronize()
    fresh.close()


def test_cli_gltf_in_frame_out(tmp_path):
    """The reference's CLI shape on a glTF file: culling -> rasteriser -> clusters -> opaque -> mips -> transmission
    -> tonemap -> PNG, all on the GPU."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_demo_gltf
    from transmission_renderer_amd import cli
    from transmission_renderer_amd.png import read_png
    glb, out = str(tmp_path / "demo.glb"), str(tmp_path / "frame.png")
    make_demo_gltf.main(glb)
    assert cli.main([glb, "--width", "320", "--height", "180", "--scale", "1.0", "--out", out,
                     "--hdr-out", str(tmp_path / "hdr.npy")]) == 0
    img = read_png(out)
    assert img.shape == (180, 320, 4)
    hdr = np.load(str(tmp_path / "hdr.npy")).astype(np.float32)
    covered = hdr[..., :3].sum(axis=2) > 0
    assert 0.2 < covered.mean() < 0.9 and np.isfinite(hdr).all()
    assert len(np.unique(img.reshape(-1, 4), axis=0)) > 500          # an actual picture
    assert cli.main(["NoSuchModel"]) == 2


def test_record_frame_equals_the_stepwise_sequence(renderer, ggx_lut):
    """tr_record_frame (one native call per frame) reproduces, bit for bit, the frame the individual entry points
    give when called in the reference's order; a second frame from another camera re-uses every buffer."""
    from transmission_renderer_amd.renderer import OpaquePyramid
    r = renderer
    w, h = 320, 180
    geo = meshes.make_mesh_scene()
    for k, view in enumerate((wire.default_camera()[1], wire.look_at_rh((3.5, 2.0, -6.0), (0.0, 1.2, -3.0), (0.0, 1.0, 0.0)))):
        sc = _scene(w, h, view)
        sc["lights"] = synthetic.make_lights(2) + [wire.Light.new_point((0.5, 1.5, -2.5), (0.2, 0.9, 0.4), 0.3)]
        q = wire.view_rotation_inverse(view)
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
        r.upload_materials(sc["materials"])
        r.upload_textures(sc["textures"])
        r.upload_lights(sc["lights"])
        r.upload_geometry(geo)
        aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
        if k == 0:
            work = r.new_frame_buffers(w, h)
        hdr, ldr = r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work)
        torch.cuda.synchronize()
        got_hdr, got_ldr = hdr.clone(), ldr.clone()
        # the same frame, one entry point at a time
        r.assign_lights_to_clusters(view, q, aabbs)
        o, t = r.new_layer(w, h), r.new_layer(w, h)
        r.draw_scene(culling, sc["push"], o, t)
        pyr = OpaquePyramid(w, h, r.device)
        hdr2 = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
        r.record(o, t, sc["uniforms"], sc["push"], hdr2, pyr)
        ldr2 = r.tonemap(hdr2)
        torch.cuda.synchronize()
        assert torch.equal(got_hdr.view(torch.int16), hdr2.view(torch.int16)) and torch.equal(got_ldr, ldr2)
        assert (got_hdr[..., :3].float().sum(dim=2) > 0).float().mean().item() > 0.2


def test_fused_front_end_with_many_culling_blocks_over_many_frames(renderer, ggx_lut):
    """The frame recorder's front launch hands the culling counts to the demultiplex behind its LAST culling workgroup (a
    ticket of relaxed agent-scope atomics, no fence: csrc/tr_shade.hip frame_front_kernel).  A scene of 20 000 small
    instances — 79 culling workgroups spread over every XCD, half of them outside the frustum — through 40 consecutive frames
    from two alternating cameras: every frame bit for bit the stepwise frame of its camera (whose culling, demultiplex and
    scans are launches of their own)."""
    r = renderer
    w, h = 320, 180
    rng = np.random.default_rng(7)
    S = meshes.Similarity
    mb = meshes.ModelBuffers()
    mb.add_primitive(meshes.plane(30.0, 30.0, cells=4), 0, [(S(np.array([0, 0.6, -6.0], np.float32)), 3)])
    small = meshes.box(0.5, 0.5, 0.5)

    def scatter(n, material):
        return [(S(np.array([rng.uniform(-30, 30), rng.uniform(0.7, 6), rng.uniform(-30, 12)], np.float32), rng.uniform(0.02, 0.08)),
                 material(k)) for k in range(n)]
    mb.add_primitive(small, 0, scatter(15_000, lambda k: k % 16))
    mb.add_primitive(small, 2, scatter(5_000, lambda k: 4))
    geo = mb.finish()
    assert len(geo["instances"]) > 64 * 256
    views = (wire.default_camera()[1], wire.look_at_rh((3.5, 2.0, -6.0), (0.0, 1.2, -3.0), (0.0, 1.0, 0.0)))
    sc = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
    r.upload_materials(sc["materials"])
    r.upload_lights(sc["lights"])
    r.upload_geometry(geo)
    aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
    cams = [(v, wire.view_rotation_inverse(v), wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), v)) for v in views]
    want = [_stepwise_frame(r, sc, c, v, q, aabbs, w, h) for v, q, c in cams]
    assert not torch.equal(want[0], want[1])
    work = r.new_frame_buffers(w, h)
    for k in range(40):
        v, q, c = cams[k & 1]
        hdr, _ = r.record_frame(sc["uniforms"], sc["push"], c, v, q, aabbs, work)
        if k % 5 == 4 or k < 2:   # (frames in between run back to back: nothing drains the GPU between their front launches)
            torch.cuda.synchronize()
            assert torch.equal(hdr.view(torch.int16), want[k & 1].view(torch.int16)), k
    torch.cuda.synchronize()
    assert torch.equal(hdr.view(torch.int16), want[1].view(torch.int16))


def test_record_frame_replays_from_a_hip_graph(renderer, ggx_lut):
    """A whole frame — culling, set-up with its look-back scan, rasteriser, both shading launches, mips — captured into a HIP
    graph once a frame has run outside the capture, and replayed: every replay is the direct frame bit for bit (the scan's
    frame counter lives on the device and moves on with every replay; a repeated tag would make stale look-back words read as
    current).  What cannot be captured is refused: a frame of another size would have to re-lay the visibility buffers out."""
    r = renderer
    w, h = 320, 180
    geo = meshes.make_mesh_scene()
    view = wire.default_camera()[1]
    sc = _scene(w, h, view)
    q = wire.view_rotation_inverse(view)
    culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    r.upload_materials(sc["materials"])
    r.upload_textures(sc["textures"])
    r.upload_lights(sc["lights"])
    r.upload_geometry(geo)
    aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
    work = r.new_frame_buffers(w, h)
    frame = lambda: r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work)
    hdr, ldr = frame()
    torch.cuda.synchronize()
    want_hdr, want_ldr = hdr.clone(), ldr.clone()
    assert (want_hdr[..., :3].float().sum(dim=2) > 0).float().mean().item() > 0.2
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        frame()                      # (once on the capturing stream: it has then seen every table)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        frame()
    for k in range(4):
        work["hdr"].zero_()
        work["ldr"].zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(work["hdr"].view(torch.int16), want_hdr.view(torch.int16)) and torch.equal(work["ldr"], want_ldr), k
    # a direct frame after the replays starts from what they left behind
    hdr, ldr = frame()
    torch.cuda.synchronize()
    assert torch.equal(hdr.view(torch.int16), want_hdr.view(torch.int16)) and torch.equal(ldr, want_ldr)
    # another frame size inside a capture: refused before anything is enqueued
    from transmission_renderer_amd._lib import TrError
    small = r.new_frame_buffers(160, 96)
    sc2 = _scene(160, 96, view)
    aabbs2 = r.write_cluster_data(sc2["uniforms"], wire.inverse_perspective(160, 96), (160, 96))
    culling2 = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(160, 96), view)
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(TrError):
        with torch.cuda.graph(g2, stream=side):
            r.record_frame(sc2["uniforms"], sc2["push"], culling2, view, q, aabbs2, small)
    torch.cuda.synchronize()
    frame()
    torch.cuda.synchronize()


def _stepwise_frame(r, sc, culling, view, q, aabbs, w, h, dtype=torch.float16):
    from transmission_renderer_amd.renderer import OpaquePyramid
    r.assign_lights_to_clusters(view, q, aabbs)
    o, t = r.new_layer(w, h), r.new_layer(w, h)
    r.draw_scene(culling, sc["push"], o, t)
    pyr = OpaquePyramid(w, h, r.device)
    hdr = torch.zeros((h, w, 4), dtype=dtype, device=r.device)
    r.record(o, t, sc["uniforms"], sc["push"], hdr, pyr)
    torch.cuda.synchronize()
    return hdr


def test_record_frame_shades_from_visibility_words_mixed_classes(renderer, ggx_lut, tmp_path):
    """The frame recorder shades RGBA16F frames straight from the rasteriser's visibility words (no resolve, no planes):
    bit for bit the stepwise sequence through the TGB-v1 planes, on a glTF scene that mixes untextured, lite-class,
    full-class and transmissive materials; frame sizes change between frames (the visibility buffers are re-laid out
    and must come back clean), the camera moves, and an RGBA32F frame goes through the planes."""
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1] / "tools"))
    import make_demo_gltf
    from transmission_renderer_amd import gltf
    r = renderer
    glb = str(tmp_path / "demo.glb")
    make_demo_gltf.main(glb)
    loaded = gltf.load_gltf(glb, base_transform=meshes.Similarity(np.array([0.0, 2.0, 0.0], np.float32), 1.0))
    geo = loaded.geometry()
    flags = []
    cams = (wire.default_camera()[1], wire.look_at_rh((3.5, 2.5, -7.0), (0.0, 1.5, -3.0), (0.0, 1.0, 0.0)),
            wire.look_at_rh((-2.5, 3.0, -6.5), (0.5, 1.8, -3.0), (0.0, 1.0, 0.0)))
    work = {}
    for k, (view, (w, h), dtype) in enumerate(zip(cams + cams[:1], ((640, 360), (483, 273), (640, 360), (640, 360)),
                                                  (torch.float16, torch.float16, torch.float16, torch.float32))):
        sc = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
        sc["materials"] = loaded.materials
        sc["textures"] = loaded.textures
        eye = np.linalg.inv(np.asarray(view, np.float64).T)[:3, 3]
        sc["push"] = wire.make_push_constants(w, h, eye=eye.astype(np.float32), view=view)
        q = wire.view_rotation_inverse(view)
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
        r.upload_materials(sc["materials"])
        r.upload_textures(sc["textures"])
        r.upload_lights(sc["lights"])
        r.upload_geometry(geo)
        aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
        if (w, h) not in work:
            work[(w, h)] = r.new_frame_buffers(w, h)
        wk = dict(work[(w, h)])
        if dtype == torch.float32:
            wk["hdr"] = torch.zeros((h, w, 4), dtype=torch.float32, device=r.device)
        for _ in range(2):   # (the second frame of each pair starts from what the first left behind)
            hdr, _ldr = r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, wk, tonemap=dtype == torch.float16)
            torch.cuda.synchronize()
        want = _stepwise_frame(r, sc, culling, view, q, aabbs, w, h, dtype)
        bits = torch.int16 if dtype == torch.float16 else torch.int32
        assert torch.equal(hdr.view(bits), want.view(bits)), f"frame {k} ({w}x{h}, {dtype})"
        if dtype == torch.float16:   # (the recorder's tonemap stores the clear colour's value into untouched tiles)
            assert torch.equal(_ldr, r.tonemap(want)), f"tonemapped frame {k} ({w}x{h})"
        assert (hdr[..., :3].float().sum(dim=2) > 0).float().mean().item() > 0.2
        flags.append(k)
    assert len(flags) == 4


def test_record_frame_from_visibility_words_untextured(ggx_lut):
    """The same for a scene without a single texture slot (the kernels the headline bench runs, fed from visibility words)."""
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(ggx_lut)
    w, h = 514, 290
    geo = meshes.make_mesh_scene(extra_instances=True)
    for view in (wire.default_camera()[1], wire.look_at_rh((3.5, 2.0, -6.0), (0.0, 1.2, -3.0), (0.0, 1.0, 0.0))):
        sc = synthetic.make_scene(w, h, num_point_lights=3, textured=False, with_gbuffer=False)
        eye = np.linalg.inv(np.asarray(view, np.float64).T)[:3, 3]
        sc["push"] = wire.make_push_constants(w, h, eye=eye.astype(np.float32), view=view)
        q = wire.view_rotation_inverse(view)
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
        r.upload_materials(sc["materials"])
        r.upload_lights(sc["lights"])
        r.upload_geometry(geo)
        aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
        work = r.new_frame_buffers(w, h)
        for _ in range(2):
            hdr, _ldr = r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work)
            torch.cuda.synchronize()
        want = _stepwise_frame(r, sc, culling, view, q, aabbs, w, h)
        assert torch.equal(hdr.view(torch.int16), want.view(torch.int16))
        assert (hdr[..., :3].float().sum(dim=2) > 0).float().mean().item() > 0.2


@pytest.mark.parametrize("case,w,h", [("everything culled", 514, 290), ("no transmissive draw", 514, 290), ("no opaque draw", 514, 290),
                                      ("one sliver", 514, 290), ("everything culled", 3442, 1442), ("one sliver", 3442, 1442)])
def test_record_frame_of_degenerate_scenes(ggx_lut, case, w, h):
    """Frames whose layers are (nearly) empty — the camera looks away from every instance (each pixel is the clear colour: the
    opaque launch's background path writes the target, pyramid levels 0 and 1 and the presented frame from constants), a
    scene without a transmissive draw (the list of covered tiles stays empty), one without an opaque draw (the transmissive
    launch shades over the clear colour), one long sliver triangle — through tr_record_frame twice, bit for bit the stepwise
    sequence, the whole pyramid bit for bit tr_generate_mips on the frame's level 0.  3442x1442: a frame large enough for a wave to
    meet several background tiles (above ~3 Mpixels) whose width is no multiple of 16 and whose height no multiple of 4 — the lanes
    that were outside the rect on a wave's first background tile must present the clear texel, not 0, on its later ones."""
    from transmission_renderer_amd.renderer import OpaquePyramid, TransmissionRenderer
    S = meshes.Similarity
    view = wire.default_camera()[1]
    mb = meshes.ModelBuffers()
    at = lambda x, y, z, s=1.0: S(np.array([x, y, z], np.float32), s)   # noqa: E731
    if case == "everything culled":
        mb.add_primitive(meshes.uv_sphere(1.0, 12, 6), 0, [(at(0.0, 3.0, 30.0), 1), (at(-60.0, 1.0, -3.0), 6)])
        mb.add_primitive(meshes.box(0.5, 0.5, 0.5), 2, [(at(0.0, 80.0, -3.0), 4)])
    elif case == "no transmissive draw":
        mb.add_primitive(meshes.plane(8.0, 8.0, cells=4, uv_repeat=4.0), 0, [(at(0, 0.6, -3.0), 3)])
        mb.add_primitive(meshes.uv_sphere(1.0, 16, 8), 0, [(at(-0.9, 1.6, -2.6, 0.55), 1), (at(1.1, 1.4, -3.4, 0.7), 6)])
    elif case == "no opaque draw":
        mb.add_primitive(meshes.uv_sphere(1.0, 16, 8), 2, [(at(0.15, 1.7, -1.9, 0.6), 4), (at(-1.6, 1.2, -3.9, 0.45), 10)])
        mb.add_primitive(meshes.box(0.7, 0.4, 0.08), 2, [(at(1.3, 1.0, -2.2), 14)])
    else:
        # (about two pixels tall at its thick end and eight world units long; both windings: one of the two faces the camera)
        sliver = meshes.Mesh(position=np.array([[-4.0, 1.0, -3.0], [4.0, 1.0, -3.0], [4.0, 1.03, -3.0]], np.float32),
                             normal=np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (3, 1)),
                             uv=np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0]], np.float32), index=np.array([0, 1, 2, 0, 2, 1], np.uint32))
        mb.add_primitive(sliver, 0, [(at(0.0, 0.5, 0.0), 5)])
        mb.add_primitive(sliver, 2, [(at(0.0, 0.0, 0.3), 4)])
    geo = mb.finish()
    r = TransmissionRenderer(0)
    try:
        r.upload_ggx_lut(ggx_lut)
        sc = _scene(w, h, view)
        q = wire.view_rotation_inverse(view)
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
        r.upload_materials(sc["materials"])
        r.upload_textures(sc["textures"])
        r.upload_lights(sc["lights"])
        r.upload_geometry(geo)
        aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
        work = r.new_frame_buffers(w, h)
        for _ in range(2):
            work["pyramid"].texels[w * h:].fill_(float("nan"))
            hdr, ldr = r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work)
            torch.cuda.synchronize()
        want = _stepwise_frame(r, sc, culling, view, q, aabbs, w, h)
        assert torch.equal(hdr.view(torch.int16), want.view(torch.int16)), case
        assert torch.equal(ldr, r.tonemap(want)), case
        covered = (hdr[..., :3].float().sum(dim=2) > 0).float().mean().item()
        if case == "everything culled":
            assert covered == 0.0 and torch.equal(hdr[..., 3].view(torch.int16), torch.full((h, w), 0x3C00, dtype=torch.int16, device=hdr.device))
        else:
            assert covered > (0.0005 if case == "one sliver" else 0.02), (case, covered)
        ref = OpaquePyramid(w, h, r.device)
        ref.level(0).copy_(work["pyramid"].level(0))
        r.generate_mips(ref)
        torch.cuda.synchronize()
        for l in range(ref.levels):
            assert torch.equal(work["pyramid"].level(l).view(torch.int16), ref.level(l).view(torch.int16)), (case, l)
    finally:
        r.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("w,h", [(320, 180), (483, 273), (514, 290), (1920, 1080), (1024, 1024), (3840, 2160)])
def test_record_frame_leaves_the_whole_opaque_pyramid(ggx_lut, w, h):
    """The pyramid a recorded frame leaves behind — level 1 written by the opaque launch itself from its wave tiles' quads when
    both frame sizes are even, the chain's launches from level 2 on; from level 1 on for odd sizes — is, level by level and bit
    for bit, tr_generate_mips on the same level 0; over three consecutive frames, levels 1.. overwritten with NaN in between
    (nothing survives from the frame before)."""
    from transmission_renderer_amd.renderer import OpaquePyramid, TransmissionRenderer
    r = TransmissionRenderer(0)
    try:
        r.upload_ggx_lut(ggx_lut)
        geo = meshes.make_mesh_scene(extra_instances=True)
        view = wire.default_camera()[1]
        sc = _scene(w, h, view)
        q = wire.view_rotation_inverse(view)
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
        r.upload_materials(sc["materials"])
        r.upload_textures(sc["textures"])
        r.upload_lights(sc["lights"])
        r.upload_geometry(geo)
        aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
        work = r.new_frame_buffers(w, h)
        ref = OpaquePyramid(w, h, r.device)
        for frame in range(3):
            work["pyramid"].texels[w * h:].fill_(float("nan"))
            r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work)
            torch.cuda.synchronize()
            got = work["pyramid"]
            ref.level(0).copy_(got.level(0))
            r.generate_mips(ref)
            torch.cuda.synchronize()
            for l in range(got.levels):
                assert torch.equal(got.level(l).view(torch.int16), ref.level(l).view(torch.int16)), (frame, l)
            assert torch.isfinite(got.level(got.levels - 1).float()).all()
    finally:
        r.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("w,h", [(3840, 2160), (7680, 4320)])
def test_record_frame_equals_the_stepwise_sequence_at_4k_and_8k(ggx_lut, tmp_path, w, h):
    """BASELINE configs 4 / 5's frame sizes through the frame recorder (culling -> rasteriser -> visibility-word shading ->
    mips -> transmissive -> tonemap in one native call) on the demo glTF: bit for bit the stepwise sequence through the
    TGB-v1 planes (each step of which the other tests hold against the oracle)."""
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1] / "tools"))
    import make_demo_gltf
    from transmission_renderer_amd import gltf
    from transmission_renderer_amd.renderer import OpaquePyramid, TransmissionRenderer
    glb = str(tmp_path / "demo.glb")
    make_demo_gltf.main(glb)
    loaded = gltf.load_gltf(glb, base_transform=meshes.Similarity(np.array([0.0, 2.0, 0.0], np.float32), 1.0))
    geo = loaded.geometry()
    r = TransmissionRenderer(0)
    try:
        r.upload_ggx_lut(ggx_lut)
        view = wire.default_camera()[1]
        sc = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
        sc["materials"], sc["textures"] = loaded.materials, loaded.textures
        q = wire.view_rotation_inverse(view)
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
        r.upload_materials(sc["materials"])
        r.upload_textures(sc["textures"])
        r.upload_lights(sc["lights"])
        r.upload_geometry(geo)
        aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
        work = r.new_frame_buffers(w, h)
        for _ in range(2):      # (the second frame starts from what the first left behind)
            hdr, _ldr = r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work)
            torch.cuda.synchronize()
        got = hdr.clone()
        del work, hdr, _ldr
        torch.cuda.empty_cache()
        want = _stepwise_frame(r, sc, culling, view, q, aabbs, w, h, torch.float16)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
        assert (got[..., :3].float().sum(dim=2) > 0).float().mean().item() > 0.2
    finally:
        r.close()
        torch.cuda.empty_cache()


def test_two_contexts_with_frames_in_flight_are_independent(ggx_lut):
    """Two contexts (each its own work buffers) recording frames on two HIP streams at once — how a renderer keeps two
    frames in flight (bench.py's frame_pipeline.two_frames_in_flight): each context's frames are bit for bit the frames
    it renders alone, i.e. the library shares no mutable state between contexts."""
    from transmission_renderer_amd.renderer import TransmissionRenderer
    w, h = 514, 290
    geo = meshes.make_mesh_scene(extra_instances=True)
    views = (wire.default_camera()[1], wire.look_at_rh((3.5, 2.0, -6.0), (0.0, 1.2, -3.0), (0.0, 1.0, 0.0)))
    ctxs = []
    for view in views:                       # (a different camera per context: a mix-up cannot go unnoticed)
        r = TransmissionRenderer(0)
        r.upload_ggx_lut(ggx_lut)
        sc = _scene(w, h, view)
        r.upload_materials(sc["materials"])
        r.upload_textures(sc["textures"])
        r.upload_lights(sc["lights"])
        r.upload_geometry(geo)
        q = wire.view_rotation_inverse(view)
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
        aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
        work = r.new_frame_buffers(w, h)
        frame = (lambda r=r, sc=sc, culling=culling, view=view, q=q, aabbs=aabbs, work=work:
                 r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work))
        hdr, ldr = frame()
        torch.cuda.synchronize()
        ctxs.append((r, frame, hdr.clone(), ldr.clone()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(3):
        outs = []
        for k in range(8):
            with torch.cuda.stream(streams[k % 2]):
                outs.append(ctxs[k % 2][1]())
        torch.cuda.synchronize()
        for k in (6, 7):                     # (a context's targets are its own: the last frame of each is what they hold)
            hdr, ldr = outs[k]
            assert torch.equal(hdr.view(torch.int16), ctxs[k % 2][2].view(torch.int16))
            assert torch.equal(ldr, ctxs[k % 2][3])
    assert not torch.equal(ctxs[0][2], ctxs[1][2])
    for c in ctxs:
        c[0].close()
