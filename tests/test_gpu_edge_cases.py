"""Edge cases of the reference's formulas that the synthetic scenes (ior in [1, 2], front-facing normals, thin volumes) never
reach, GPU against the pinned fp32 oracle (run with -m gpu):

  * ior < 1 at grazing angles: `refract` has no total-internal-reflection guard (glam-pbr/src/lib.rs:248-256): k < 0 makes
    sqrt(k) NaN in the reference; the kernel (which skips the re-normalisation: the refracted vector is unit length by
    Snell's law whenever it exists) must treat those pixels the same way — in both, the NaN exit point ends in the
    sampler's clamp and the pixel shades the pyramid's corner texel — and agree everywhere;
  * clip.w <= 0: a back-facing normal sends the "refracted" ray towards the camera, a thick volume carries the exit point
    behind it; uv = clip.xy / clip.w is then mirrored or infinite — undefined in the reference, defined by IEEE arithmetic and
    the clamp of the sampler in both restatements: the frames must agree;
  * v = -l (the light behind the pixel on the view ray): Halfway::new normalises a zero vector (:62-68) — NaN in the
    reference's lobe; the kernel floors |v + l|^2 (tr_kernels.h, eval_light) and stays finite: a STATED divergence, counted.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from transmission_renderer_amd import synthetic, wire  # noqa: E402
from test_gpu_parity import _norm_err, _upload_scene  # noqa: E402


@pytest.fixture(scope="module")
def renderer(ggx_lut):
    if not torch.cuda.is_available():
        pytest.fail("no HIP device: the -m gpu tests must run on the GPU box")
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(ggx_lut)
    yield r
    r.close()


def _grazing_gbuffer(scene, flip_right_half=False):
    """The synthetic planes with the normal swept from facing the camera (left edge) to 89.7 degrees off (right edge)."""
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in scene["gbuffer"].items()}
    h, w = g["material_id"].shape
    eye = np.array(scene["push"].view_position[:3], dtype=np.float64)
    v = eye - g["pos_depth"][..., :3].astype(np.float64)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    t = np.cross(v, np.array([0.0, 1.0, 0.0]))
    t /= np.linalg.norm(t, axis=-1, keepdims=True)
    theta = np.deg2rad(np.linspace(0.0, 89.7, w))[None, :, None]
    n = np.cos(theta) * v + np.sin(theta) * t
    if flip_right_half:
        n[:, w // 2:] *= -1.0
    g["nrm_scale"][..., :3] = (n * 0.8).astype(np.float32)          # un-normalised, like an interpolant
    return g


def _both(renderer, ggx_lut, scene, g, w, h):
    from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid
    r = renderer
    _upload_scene(r, scene)
    mip0 = synthetic.make_opaque_mip0(w, h)
    pyr = OpaquePyramid(w, h, r.device)
    pyr.level(0).copy_(torch.from_numpy(mip0).to(r.device))
    r.generate_mips(pyr)
    got = torch.zeros((h, w, 4), dtype=torch.float32, device=r.device)
    r.shade_transmission(GBufferPlanes.from_numpy(g, r.device), scene["uniforms"], scene["push"], pyr, got)
    torch.cuda.synchronize()
    b = oracle.SceneBinding(scene, ggx_lut)
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    want = np.zeros((h, w, 4), dtype=np.float32)
    oracle.shade_transmission(b, g, tex, hdr_f32=want, nthreads=4)
    return got.cpu().numpy(), want


def _agreement(got, want, what, max_nan_mismatch):
    gn, wn = ~np.isfinite(got[..., :3]).all(axis=2), ~np.isfinite(want[..., :3]).all(axis=2)
    mismatch = gn != wn
    both = ~gn & ~wn
    e = _norm_err(got[both], want[both])[:, :3]
    rmse = float(np.sqrt((e ** 2).mean(axis=0)).max())
    print(f"[edge] {what}: non-finite pixels gpu {int(gn.sum())} / oracle {int(wn.sum())} of {gn.size}, disagreeing {int(mismatch.sum())}; "
          f"finite pixels rmse {rmse:.2e}, max {float(np.abs(e).max()):.2e}")
    assert mismatch.sum() <= max_nan_mismatch, (what, int(mismatch.sum()))
    return rmse, float(np.abs(e).max()), int(gn.sum()), int(wn.sum())


@pytest.mark.parametrize("ior", [0.6, 0.9])
def test_ior_below_one_at_grazing_angles(renderer, ggx_lut, ior):
    w, h = 256, 96
    scene = synthetic.make_scene(w, h, num_point_lights=1)
    for m in scene["materials"]:
        m.index_of_refraction = ior
        m.transmission_factor = 1.0
        m.thickness_factor = min(m.thickness_factor, 0.3)
    g = _grazing_gbuffer(scene)
    got, want = _both(renderer, ggx_lut, scene, g, w, h)
    # total internal reflection sets in where sin(theta) > ior: k < 0 there and sqrt(k) is NaN on both sides — a NaN that
    # never reaches the frame: the exit point's uv goes through the sampler's clamp (NaN -> texel 0 on both sides, Vulkan's
    # clamp-to-edge restated the same way in the oracle and in pyramid_issue_record's v_med3), so both shade the corner texel
    theta = np.deg2rad(np.linspace(0.0, 89.7, w))
    tir = float((np.sin(theta) > ior).mean())
    assert tir > 0.1                                       # the case is reached: a band of columns is past the critical angle
    rmse, worst, gn, wn = _agreement(got, want, f"ior {ior} ({tir:.0%} of the columns past the critical angle)", max_nan_mismatch=0)
    assert gn == 0 and wn == 0
    assert rmse <= 1e-4 and worst <= 5e-3


def test_exit_point_behind_the_camera(renderer, ggx_lut):
    w, h = 256, 96
    scene = synthetic.make_scene(w, h, num_point_lights=1)
    for m in scene["materials"]:
        m.transmission_factor = 1.0
        m.thickness_factor = 40.0          # times model_scale (0.5 ... 2): far beyond the camera, 2.6 m away
    g = _grazing_gbuffer(scene, flip_right_half=True)
    got, want = _both(renderer, ggx_lut, scene, g, w, h)
    behind = float((_exit_clip_w(scene, g) <= 0.0).mean())
    assert behind > 0.01                                   # the case is reached (hundreds of pixels)
    rmse, worst, gn, wn = _agreement(got, want, f"clip.w <= 0 on {behind:.0%} of the pixels", max_nan_mismatch=8)
    assert rmse <= 1e-4 and worst <= 5e-3


def _exit_clip_w(scene, g):
    """clip.w of the refracted ray's exit point (glam-pbr/src/lib.rs:248-268, 328-332), restated in float64."""
    mats = scene["materials"]
    pos = g["pos_depth"][..., :3].astype(np.float64)
    n = g["nrm_scale"][..., :3].astype(np.float64)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    scale = g["nrm_scale"][..., 3].astype(np.float64)
    mid = g["material_id"]
    eta = 1.0 / np.array([m.index_of_refraction for m in mats])[mid]
    thick = np.array([m.thickness_factor for m in mats])[mid]
    eye = np.array(scene["push"].view_position[:3], dtype=np.float64)
    P = np.array(scene["push"].proj_view, dtype=np.float64).reshape(4, 4).T
    v = eye - pos
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    nov = (n * v).sum(-1)
    k = 1.0 - eta ** 2 * (1.0 - nov ** 2)
    cn = -eta * nov + np.sqrt(np.maximum(k, 0.0))
    ex = pos + (-eta[..., None] * v - cn[..., None] * n) * (thick * scale)[..., None]
    return ex @ P[3, :3] + P[3, 3]


def test_light_exactly_behind_the_pixel_on_the_view_ray(renderer, ggx_lut):
    """v = -l: the reference's Halfway::new normalises the zero vector.  One light is placed on the view ray of the frame's
    centre pixel, behind the surface; that pixel is the stated divergence (oracle NaN or huge, kernel finite), every
    other pixel agrees."""
    w, h = 64, 32
    scene = synthetic.make_scene(w, h, num_point_lights=1)
    g = scene["gbuffer"]
    eye = np.array(scene["push"].view_position[:3], dtype=np.float32)
    p = g["pos_depth"][h // 2, w // 2, :3]
    behind = p + (p - eye) * np.float32(0.5)
    scene["lights"][0].position_and_spotlight_epsilon[0] = float(behind[0])
    scene["lights"][0].position_and_spotlight_epsilon[1] = float(behind[1])
    scene["lights"][0].position_and_spotlight_epsilon[2] = float(behind[2])
    got, want = _both(renderer, ggx_lut, scene, g, w, h)
    assert np.isfinite(got).all()          # the kernel's floor on |v + l|^2 keeps the lobe finite
    gn, wn = ~np.isfinite(got[..., :3]).all(axis=2), ~np.isfinite(want[..., :3]).all(axis=2)
    e = np.abs(_norm_err(got, want)[..., :3]).max(axis=2)
    e = np.where(wn, np.inf, e)
    diverging = e > 1e-3
    print(f"[edge] v = -l: oracle non-finite pixels {int(wn.sum())}, pixels diverging by more than 1e-3: {int(diverging.sum())} "
          f"(centre pixel: gpu {got[h // 2, w // 2, :3]}, oracle {want[h // 2, w // 2, :3]})")
    ys, xs = np.nonzero(diverging)
    assert len(ys) <= 9 and (len(ys) == 0 or (np.abs(ys - h // 2).max() <= 1 and np.abs(xs - w // 2).max() <= 1))
