"""GPU parity of the batched glam-pbr API (tr_basic_brdf & co through the C ABI) against the oracle.

Criteria, as for the frames (tests/test_gpu_parity.py): errors are normalised by max(|reference|, 1);
  * basic_brdf / transmission_btdf / ibl_volume_refraction run the passes' well-conditioned device code and are
    compared with the fp64 evaluation of the reference's formulas (RMSE <= 1e-4, max <= 5e-3 over all elements) and
    with the fp32 oracle where that is itself within 1e-5 of fp64;
  * the small functions are the reference's formulas in its operation order: <= 4 ulp of the fp32 oracle;
  * composing the API calls like `fragment_transmission` does reproduces the transmissive pass's pixels.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from tests import glam_cases  # noqa: E402
from transmission_renderer_amd import synthetic, wire  # noqa: E402


@pytest.fixture(scope="module")
def api(ggx_lut):
    if not torch.cuda.is_available():
        pytest.fail("no HIP device: the -m gpu tests must run on the GPU box")
    from transmission_renderer_amd.glam_pbr import GlamPbr
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(ggx_lut)
    yield GlamPbr(r)
    r.close()


def _nerr(got, ref):
    return (got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)


def _check(got, o32, o64, what, well_conditioned_fraction):
    assert np.isfinite(got).all(), what
    e64 = _nerr(got, o64)
    assert np.sqrt((e64 ** 2).mean(axis=0)).max() <= 1e-4, (what, "rmse vs fp64", np.sqrt((e64 ** 2).mean(axis=0)))
    assert np.abs(e64).max() <= 5e-3, (what, "max vs fp64", np.abs(e64).max())
    noise = np.abs(_nerr(o32, o64)).max(axis=1)
    good = noise <= 1e-5
    assert good.mean() >= well_conditioned_fraction, (what, good.mean())
    e32 = _nerr(got, o32)[good]
    assert np.sqrt((e32 ** 2).mean(axis=0)).max() <= 1e-5, (what, "rmse vs fp32 (well conditioned)")
    bad = ~good   # where the reference's fp32 formulas are noise, the device code is nearer to the exact value
    if bad.any():
        over = np.abs(e64[bad]).max(axis=1) - noise[bad]
        print(what, "ill-conditioned elements:", int(bad.sum()), "worst excess over the fp32 oracle's own noise:", over.max())
        assert over.max() <= 3e-4, (what, over.max(), (over > 1e-4).sum())   # measured: 1.3e-4 on 1 of 200 000


def test_basic_brdf(api):
    p = glam_cases.basic_brdf_params(200_000)
    got = api.basic_brdf(p).cpu().numpy()
    _check(got, oracle.basic_brdf_batch(p), oracle.basic_brdf_batch(p, fp64=True), "basic_brdf", 0.97)
    # device-resident records give the same bits as uploaded ones
    dev = torch.from_numpy(p.view(np.float32).reshape(-1, 22).copy()).cuda()
    assert torch.equal(api.basic_brdf(dev).cpu(), torch.from_numpy(got))


def test_transmission_btdf(api):
    p = glam_cases.transmission_btdf_params(200_000)
    got = api.transmission_btdf(p).cpu().numpy()
    _check(got, oracle.transmission_btdf_batch(p), oracle.transmission_btdf_batch(p, fp64=True), "transmission_btdf", 0.95)
    q = p[:64].copy()
    q["material_params"]["index_of_refraction"] = 1.0   # alpha_t = 0: the lobe vanishes
    assert (api.transmission_btdf(q).cpu().numpy() == 0).all()


def test_ibl_volume_refraction(api, ggx_lut):
    from transmission_renderer_amd.renderer import OpaquePyramid
    w, h = 320, 200
    mip0 = synthetic.make_opaque_mip0(w, h)
    texels = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, texels)
    pyr = OpaquePyramid(w, h, api.r.device)
    pyr.texels.copy_(torch.from_numpy(texels.reshape(-1, 4)).to(api.r.device))   # the oracle's own chain: the taps see identical texels
    p = glam_cases.ibl_params(50_000, w, h)
    got = api.ibl_volume_refraction(p, pyr).cpu().numpy()
    o32 = oracle.ibl_volume_refraction_batch(p, w, h, texels, ggx_lut)
    o64 = oracle.ibl_volume_refraction_batch(p, w, h, texels, ggx_lut, fp64=True)
    ok = np.isfinite(o64).all(axis=1)   # (clip.w <= 0 is undefined in the reference)
    assert ok.mean() > 0.99
    _check(got[ok], o32[ok], o64[ok], "ibl_volume_refraction", 0.90)
    # a count that is not a multiple of the wave size, and a single element
    for n in (1, 67):
        np.testing.assert_array_equal(api.ibl_volume_refraction(p[:n], pyr).cpu().numpy(), got[:n])


def test_ibl_volume_refraction_with_the_callers_closures(api, ggx_lut):
    """ibl_volume_refraction<FSamp, GSamp> (:292-299) as its two halves.  (1) The requests against an fp64 restatement of
    :248-268 / :326-341; (2) with closures that return constants, the resolve equals the fused function run on a
    constant-colour pyramid and a constant LUT (the pinned path) — the two must agree to rounding; (3) arbitrary closures
    against an fp64 restatement of :338-353."""
    from transmission_renderer_amd.renderer import OpaquePyramid
    w, h = 320, 200
    p = glam_cases.ibl_params(20_000, w, h)
    dev = api.r.device
    seen = {}

    def fs(uv, lod):
        seen["uv"], seen["lod"] = uv.cpu().numpy().astype(np.float64), lod.cpu().numpy().astype(np.float64)
        return torch.stack([0.5 + 0.5 * torch.sin(7.0 * uv[:, 0]), 0.5 + 0.5 * torch.cos(5.0 * uv[:, 1]), 0.1 * lod + 0.2], dim=1)

    def gs(nov, rough):
        seen["nov"], seen["rough"] = nov.cpu().numpy().astype(np.float64), rough.cpu().numpy().astype(np.float64)
        return torch.stack([0.9 - 0.5 * rough, 0.05 + 0.1 * nov.abs()], dim=1)

    got = api.ibl_volume_refraction_with(p, fs, gs).cpu().numpy().astype(np.float64)
    # (1) the requests
    mp = p["material_params"]
    n, v = p["normal"].astype(np.float64), p["view"].astype(np.float64)
    ior = mp["index_of_refraction"].astype(np.float64)
    eta = 1.0 / ior
    ndi = -(n * v).sum(1)                                            # normal . incident, incident = -view
    k = 1.0 - eta ** 2 * (1.0 - ndi ** 2)
    ok = k > 1e-6
    refr = eta[:, None] * -v - (eta * ndi + np.sqrt(np.where(ok, k, 1.0)))[:, None] * n
    refr /= np.linalg.norm(refr, axis=1, keepdims=True)
    length = (p["thickness"] * p["model_scale"]).astype(np.float64)
    exit_point = p["position"].astype(np.float64) + refr * length[:, None]
    M = p["proj_view_matrix"].astype(np.float64).reshape(-1, 4, 4)   # column-major: M[:, col, row]
    clip = np.einsum("ncr,nc->nr", M, np.concatenate([exit_point, np.ones((len(p), 1))], axis=1))
    ok &= np.abs(clip[:, 3]) > 1e-3
    uv = (clip[:, :2] / clip[:, 3:4] + 1.0) / 2.0
    lod = np.log2(p["framebuffer_size_x"].astype(np.float64)) * mp["perceptual_roughness"] * np.clip(ior * 2.0 - 2.0, 0.0, 1.0)
    assert ok.mean() > 0.95
    scale = np.maximum(np.abs(uv[ok]), 1.0)
    assert (np.abs(seen["uv"][ok] - uv[ok]) / scale).max() <= 2e-4 and np.sqrt((((seen["uv"][ok] - uv[ok]) / scale) ** 2).mean()) <= 2e-5
    assert np.abs(seen["lod"] - lod).max() <= 1e-5 * max(1.0, lod.max())
    assert np.abs(seen["nov"] - (n * v).sum(1)).max() <= 1e-6 and np.abs(seen["rough"] - mp["perceptual_roughness"]).max() == 0.0
    # (3) the rest of the function in fp64 from the closures' own answers
    rgb = fs(torch.from_numpy(seen["uv"]).to(dev), torch.from_numpy(seen["lod"]).to(dev)).cpu().numpy().astype(np.float64)
    ab = gs(torch.from_numpy(seen["nov"]).to(dev), torch.from_numpy(seen["rough"]).to(dev)).cpu().numpy().astype(np.float64)
    finite = np.isfinite(p["attenuation_distance"])
    coeff = np.where(finite[:, None], -np.log(p["attenuation_colour"].astype(np.float64)) / np.where(finite, p["attenuation_distance"], 1.0)[:, None], 0.0)
    attenuated = np.exp(-coeff * length[:, None]) * rgb
    base, spec_c = mp["diffuse_colour"].astype(np.float64), mp["specular_colour"].astype(np.float64)
    metallic, spec_f = mp["metallic"].astype(np.float64)[:, None], mp["specular_factor"].astype(np.float64)[:, None]
    f0d = (((ior - 1.0) / (ior + 1.0)) ** 2)[:, None]
    dielectric = f0d * spec_c * spec_f
    f0 = dielectric + (base - dielectric) * metallic                 # calculate_combined_f0 :425-431
    f90 = spec_f + (1.0 - spec_f) * metallic                         # calculate_combined_f90 :433-435
    want = (1.0 - (f0 * ab[:, 0:1] + f90 * ab[:, 1:2])) * attenuated * base
    assert (np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max() <= 1e-5
    # (2) constant closures == the fused function on a constant pyramid and a constant LUT
    colour = np.array([0.75, 0.5, 0.25, 1.0], dtype=np.float16)
    pyr = OpaquePyramid(w, h, dev)
    pyr.texels.copy_(torch.from_numpy(np.broadcast_to(colour, (pyr.texels.shape[0], 4)).copy()).to(dev))
    lut = np.zeros_like(ggx_lut)
    lut[..., 0], lut[..., 1], lut[..., 3] = 204, 51, 255            # A = 0.8, B = 0.2 in every texel
    api.r.upload_ggx_lut(lut)
    try:
        fused = api.ibl_volume_refraction(p, pyr).cpu().numpy()
        const = api.ibl_volume_refraction_with(
            p, lambda uv, lod: torch.tensor([0.75, 0.5, 0.25], device=dev).expand(uv.shape[0], 3),
            lambda nov, rough: torch.tensor([204.0 / 255.0, 51.0 / 255.0], device=dev).expand(nov.shape[0], 2)).cpu().numpy()
    finally:
        api.r.upload_ggx_lut(ggx_lut)
    both = np.isfinite(fused).all(axis=1)
    assert both.mean() > 0.99
    assert (np.abs(fused[both] - const[both]) / np.maximum(np.abs(fused[both]), 1.0)).max() <= 2e-6


def test_small_functions(api):
    rng = np.random.default_rng(5)
    n = 100_000
    def ulps(got, ref32):
        ref32 = ref32.astype(np.float32)
        return np.abs(got.astype(np.float64) - ref32) / np.spacing(np.maximum(np.abs(ref32), np.float32(1e-30)))
    noh, nov, nol = (rng.uniform(1e-3, 1.0, n).astype(np.float32) for _ in range(3))
    rough = rng.uniform(0.0025, 1.0, n).astype(np.float32)
    assert ulps(api.d_ggx(noh, rough).cpu().numpy(), oracle.d_ggx_batch(noh, rough)).max() <= 4
    assert ulps(api.v_smith_ggx_correlated(nov, nol, rough).cpu().numpy(),
                oracle.v_smith_ggx_correlated_batch(nov, nol, rough)).max() <= 4
    voh = rng.uniform(0.0, 1.0, n).astype(np.float32)
    f0, f90 = rng.uniform(0, 1, (n, 3)).astype(np.float32), rng.uniform(0, 1, (n, 3)).astype(np.float32)
    e = api.fresnel_schlick(voh, f0, f90).cpu().numpy() - oracle.fresnel_schlick_batch(voh, f0, f90)
    assert np.abs(e).max() <= 5e-7   # x^5 by multiplication vs powf(x, 5): a few ulp of the power
    met, ior = rng.uniform(0, 1, n).astype(np.float32), rng.uniform(1, 2.5, n).astype(np.float32)
    diff = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    assert ulps(api.compute_f0(met, ior, diff).cpu().numpy(), oracle.compute_f0_batch(met, ior, diff)).max() <= 2
    fp, lp = rng.normal(size=(n, 3)).astype(np.float32) * 3, rng.normal(size=(n, 3)).astype(np.float32) * 3
    got = api.light_direction_and_attenuation(fp, lp).cpu().numpy()
    assert ulps(got, oracle.light_direction_and_attenuation_batch(fp, lp)).max() <= 2
    np.testing.assert_array_equal(api.light_direction_and_attenuation([[0, 0, 0]], [[0, 2, 0]]).cpu().numpy(), [[0, 1, 0, 2, 0.25]])
    np.testing.assert_allclose(api.d_ggx([1.0], [1.0]).cpu().numpy(), [1 / np.pi], rtol=3e-7)
    np.testing.assert_allclose(api.v_smith_ggx_correlated([1.0], [1.0], [0.3]).cpu().numpy(), [0.25], rtol=3e-7)


def test_api_composes_to_the_transmissive_pass(api, ggx_lut):
    """fragment_transmission (shader/src/lib.rs:37-162) written with the API calls — sun only — gives the pass's pixels."""
    from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid
    r = api.r
    w, h = 96, 64
    scene = synthetic.make_scene(w, h, num_point_lights=0)
    r.upload_materials(scene["materials"])
    r.upload_lights(scene["lights"] or wire.default_lights()[:1])
    counts = np.zeros_like(scene["cluster_counts"])
    r.set_cluster_tables(torch.from_numpy(counts.view(np.int32)).to(r.device),
                         torch.from_numpy(scene["light_indices"].view(np.int32)).to(r.device))
    g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
    pyr = OpaquePyramid(w, h, r.device)
    pyr.level(0).copy_(torch.from_numpy(synthetic.make_opaque_mip0(w, h)).to(r.device))
    r.generate_mips(pyr)
    hdr = torch.zeros((h, w, 4), dtype=torch.float32, device=r.device)
    r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr)
    want = hdr.cpu().numpy()[..., :3].reshape(-1, 3)

    gb = scene["gbuffer"]
    n = w * h
    mats = scene["materials"]
    mid = gb["material_id"].reshape(-1)
    pos = gb["pos_depth"].reshape(-1, 4)[:, :3]
    nrm = gb["nrm_scale"].reshape(-1, 4)
    unit = lambda v: (v / np.sqrt((v.astype(np.float32) ** 2).sum(axis=1, keepdims=True, dtype=np.float32))).astype(np.float32)
    normal = unit(nrm[:, :3])
    cam = np.array(list(scene["push"].view_position)[:3], np.float32)
    view = unit(cam[None, :] - pos)
    mp = np.zeros(n, dtype=wire.MATERIAL_PARAMS_DTYPE)
    per = lambda f: np.array([f(mats[i]) for i in mid])
    mp["diffuse_colour"] = per(lambda m: list(m.diffuse_factor)[:3])
    mp["metallic"] = per(lambda m: m.metallic_factor)
    mp["perceptual_roughness"] = per(lambda m: m.roughness_factor)
    mp["index_of_refraction"] = per(lambda m: m.index_of_refraction)
    mp["specular_colour"] = per(lambda m: list(m.specular_colour_factor))
    mp["specular_factor"] = per(lambda m: m.specular_factor)
    sun_dir = np.array(list(scene["uniforms"].sun_dir)[:3], np.float32)
    sun_int = np.array(list(scene["uniforms"].sun_intensity)[:3], np.float32)
    bp = np.zeros(n, dtype=wire.BASIC_BRDF_PARAMS_DTYPE)
    bp["normal"], bp["view"], bp["light"], bp["light_intensity"], bp["material_params"] = normal, view, sun_dir, sun_int, mp
    brdf = api.basic_brdf(bp).cpu().numpy().astype(np.float64)
    tp = np.zeros(n, dtype=wire.TRANSMISSION_BTDF_PARAMS_DTYPE)
    tp["normal"], tp["view"], tp["light"], tp["material_params"] = normal, view, sun_dir, mp
    btdf = api.transmission_btdf(tp).cpu().numpy().astype(np.float64) * sun_int[None, :]
    ip = np.zeros(n, dtype=wire.IBL_VOLUME_REFRACTION_PARAMS_DTYPE)
    ip["material_params"], ip["framebuffer_size_x"], ip["normal"], ip["view"], ip["position"] = mp, w, normal, view, pos
    ip["proj_view_matrix"] = np.array(list(scene["push"].proj_view), np.float32)[None, :]
    ip["thickness"] = per(lambda m: m.thickness_factor)
    ip["model_scale"] = nrm[:, 3]
    ip["attenuation_distance"] = per(lambda m: m.attenuation_distance)
    ip["attenuation_colour"] = per(lambda m: list(m.attenuation_colour))
    ibl = api.ibl_volume_refraction(ip, pyr).cpu().numpy().astype(np.float64)
    tf = per(lambda m: m.transmission_factor).astype(np.float64)[:, None]
    emission = per(lambda m: list(m.emissive_factor)).astype(np.float64)
    transmission = np.where(tf != 0, btdf + ibl, 0.0)   # the pass skips the term when the factor is zero
    diffuse = brdf[:, :3] + (tf * transmission - brdf[:, :3]) * tf   # lib.rs:157-159 (the factor applied twice)
    got = diffuse + brdf[:, 3:] + emission
    e = _nerr(got, want.astype(np.float64))
    # (the test normalises n and v in numpy, the pass with v_rsq_f32: ~1e-7 on the inputs, amplified on glossy pixels)
    assert np.abs(e).max() <= 5e-5 and np.sqrt((e ** 2).mean()) <= 2e-6, (np.abs(e).max(), np.sqrt((e ** 2).mean()))
