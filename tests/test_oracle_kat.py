"""Known-answer tests of the CPU oracle (oracle/tr_oracle.c) against analytic values derived from the
reference's formulas.  The reference ships no tests; these are the closed forms."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle
from transmission_renderer_amd import wire

L = oracle.load()
v3 = oracle.v3
EPS = np.finfo(np.float32).eps


def mp(diffuse=(0.8, 0.6, 0.4), metallic=0.0, rough=0.5, ior=1.5, spec_c=(1, 1, 1), spec=1.0):
    return oracle.MaterialParams(v3(diffuse), metallic, rough, ior, v3(spec_c), spec)


def test_dielectric_f0_doc_comment():
    # glam-pbr/src/lib.rs:182-195: ior 1.5 <=> 4 % reflectance
    assert abs(L.o_to_dielectric_f0(1.5) - 0.04) < 1e-7
    assert L.o_to_dielectric_f0(1.0) == 0.0


def test_d_ggx_closed_forms():
    assert abs(L.o_d_ggx(1.0, 1.0) - 1.0 / math.pi) < 1e-7            # alpha = 1: D = 1/pi for every n.h
    assert abs(L.o_d_ggx(0.3, 1.0) - 1.0 / math.pi) < 1e-7
    a = 0.25
    assert abs(L.o_d_ggx(1.0, a) - 1.0 / (math.pi * a * a)) < 1e-4   # peak = 1/(pi alpha^2)
    assert math.isnan(L.o_d_ggx(1.0, 0.0))                            # 0/0 quirk
    assert L.o_d_ggx(0.5, 0.0) == 0.0


def test_v_smith_closed_forms():
    for a in (0.0, 0.3, 1.0):
        assert abs(L.o_v_smith_ggx_correlated(1.0, 1.0, a) - 0.25) < 1e-7
    assert L.o_v_smith_ggx_correlated(0.0, 0.0, 0.0) == 0.0            # ggx == 0 -> 0 branch


def test_fresnel_schlick_endpoints():
    f0, f90 = v3((0.04, 0.1, 0.9)), v3((1, 1, 1))
    np.testing.assert_allclose(L.o_fresnel_schlick(1.0, f0, f90).np(), [0.04, 0.1, 0.9], atol=1e-7)
    np.testing.assert_allclose(L.o_fresnel_schlick(0.0, f0, f90).np(), [1, 1, 1], atol=1e-7)
    # (1 - 0.5)^5 = 1/32
    np.testing.assert_allclose(L.o_fresnel_schlick(0.5, v3((0, 0, 0)), f90).np(), [1 / 32] * 3, atol=1e-7)


def test_dot_clamps_to_epsilon_not_zero():
    assert L.o_dot_clamped(v3((1, 0, 0)), v3((-1, 0, 0))) == pytest.approx(float(EPS), rel=0, abs=0)
    assert L.o_dot_clamped(v3((1, 0, 0)), v3((0, 1, 0))) == float(EPS)
    assert L.o_dot_clamped(v3((0.6, 0.8, 0)), v3((0.6, 0.8, 0))) == pytest.approx(1.0, abs=1e-6)


def test_light_direction_and_attenuation():
    d, dist, att = oracle.Vec3(), C.c_float(), C.c_float()
    L.o_light_direction_and_attenuation(v3((0, 0, 0)), v3((0, 2, 0)), C.byref(d), C.byref(dist), C.byref(att))
    np.testing.assert_array_equal(d.np(), [0, 1, 0])
    assert dist.value == 2.0 and att.value == 0.25   # bare 1/d^2, no range window


def test_refract_normal_incidence_and_snell():
    n = v3((0, 0, 1))
    np.testing.assert_allclose(L.o_refract(v3((0, 0, -1)), n, 1.5).np(), [0, 0, -1], atol=1e-7)
    # 45 degrees into ior 1.5: sin(t) = sin(45)/1.5
    s = math.sin(math.radians(45))
    r = L.o_refract(v3((s, 0, -s)), n, 1.5).np()
    assert abs(np.linalg.norm(r) - 1.0) < 1e-6
    assert abs(r[0] - s / 1.5) < 1e-6
    # ior 1: straight through
    np.testing.assert_allclose(L.o_refract(v3((s, 0, -s)), n, 1.0).np(), [s, 0, -s], atol=1e-7)


def test_beers_law():
    light, colour = v3((2.0, 1.0, 0.5)), v3((0.5, 0.25, 0.9))
    # travelling exactly the attenuation distance multiplies by the attenuation colour
    out = L.o_apply_volume_attenuation(light, 0.7, 0.7, colour).np()
    np.testing.assert_allclose(out, [1.0, 0.25, 0.45], rtol=2e-7)
    # +INF distance is a pass-through (glam-pbr/src/lib.rs:281-282)
    np.testing.assert_array_equal(L.o_apply_volume_attenuation(light, 0.7, math.inf, colour).np(), [2.0, 1.0, 0.5])
    # monotone in path length
    a = L.o_apply_volume_attenuation(light, 0.1, 0.7, colour).np()
    b = L.o_apply_volume_attenuation(light, 0.2, 0.7, colour).np()
    assert (b < a).all()


def test_transmission_btdf_ior_one_vanishes():
    n, v, l = v3((0, 0, 1)), v3((0.6, 0, 0.8)), v3((0, 0.6, 0.8))
    np.testing.assert_array_equal(L.o_transmission_btdf(mp(ior=1.0), n, v, l).np(), [0, 0, 0])
    assert (L.o_transmission_btdf(mp(ior=1.5), n, v, l).np() > 0).all()


def test_basic_brdf_energy_and_black_metal_diffuse():
    n, v, l = v3((0, 0, 1)), v3((0, 0, 1)), v3((0, 0, 1))
    r = L.o_basic_brdf(n, l, v3((1, 1, 1)), v, mp(rough=1.0))
    # normal incidence, alpha=1: F = f0 = 0.04, D = 1/pi, V = 0.25
    np.testing.assert_allclose(r.specular.np(), [0.04 / math.pi * 0.25] * 3, rtol=1e-5)
    np.testing.assert_allclose(r.diffuse.np(), np.array([0.8, 0.6, 0.4]) * (1 - 0.04) / math.pi, rtol=1e-5)
    rm = L.o_basic_brdf(n, l, v3((1, 1, 1)), v, mp(metallic=1.0))
    np.testing.assert_array_equal(rm.diffuse.np(), [0, 0, 0])


def test_compute_f0():
    np.testing.assert_allclose(L.o_compute_f0(0.0, 1.5, v3((0.5, 0.5, 0.5))).np(), [0.04] * 3, atol=1e-7)
    np.testing.assert_allclose(L.o_compute_f0(1.0, 1.5, v3((0.5, 0.6, 0.7))).np(), [0.5, 0.6, 0.7], atol=1e-7)


def test_cluster_coefficients_and_depth_slice():
    c = wire.LightClusterCoefficients()
    L.o_light_cluster_coefficients_new(0.01, 500.0, 16, C.byref(c))
    assert abs(c.scale - 1.0250076) < 2e-6 and abs(c.bias - 6.8100033) < 2e-6
    py = wire.LightClusterCoefficients.new()
    assert py.scale == c.scale and py.bias == c.bias
    # reversed-Z depth 1 is the near plane -> slice 0; depth 0 the far plane -> slice 16 (unclamped quirk)
    assert L.o_get_depth_slice(C.byref(c), 1.0) == 0
    assert L.o_get_depth_slice(C.byref(c), 0.0) in (15, 16)   # log2(500)*scale+bias = 16 - 1 ulp in fp32
    slices = [L.o_get_depth_slice(C.byref(c), d) for d in np.linspace(1.0, 0.0, 200)]
    assert slices == sorted(slices)


def test_host_helpers():
    assert L.o_mip_levels_for_size(3840, 2160) == 12 and L.o_mip_levels_for_size(256, 256) == 9
    assert L.o_mip_levels_for_size(1920, 1080) == 11 and L.o_mip_levels_for_size(7680, 4320) == 13
    assert [wire.mip_levels_for_size(*s) for s in ((3840, 2160), (256, 256), (1920, 1080), (7680, 4320))] == [12, 9, 11, 13]
    sun = (C.c_float * 3)()
    L.o_sun_as_normal(1.1, 4.8, C.byref(sun))
    np.testing.assert_allclose(list(sun), [-0.451856, 0.891207, 0.039689], atol=2e-6)
    np.testing.assert_allclose(wire.sun_as_normal(), list(sun), atol=1e-7)
    m = (C.c_float * 16)()
    L.o_perspective_matrix_reversed(1920, 1080, C.byref(m))
    m = np.array(list(m), dtype=np.float64).reshape(4, 4)  # [column][row]
    np.testing.assert_allclose(m, wire.perspective_matrix_reversed(1920, 1080), rtol=1e-6)
    for z, want in ((-0.01, 1.0), (-500.0, 0.0)):   # view z -> reversed depth
        clip = m.T @ np.array([0, 0, z, 1.0])
        assert abs(clip[2] / clip[3] - want) < 1e-6


def test_spotlight_factor():
    spot = wire.Light.new_spot((0, 4, 0), (1, 1, 0.5), 50.0, (0, -1, 0), 0.7, 0.8)
    eps = math.cos(0.7) - math.cos(0.8)
    # light straight above the fragment, pointing down: theta = 1
    f = L.o_spotlight_factor(C.byref(spot), v3((0, 1, 0)))
    assert abs(f - (1 - math.cos(0.8)) / eps) < 1e-4
    # outside the outer cone: clamped to 0
    assert L.o_spotlight_factor(C.byref(spot), v3((1, 0, 0))) == 0.0


def test_half_conversion_matches_numpy_rtne():
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.normal(size=4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-8, 6, 4000),
                           np.array([0.0, -0.0, 65504.0, 65520.0, 1e-8, 6.1e-5, 5.96e-8, 2.98e-8, np.inf, -np.inf],
                                    dtype=np.float32)]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = np.array([L.o_f32_to_f16(float(v)) for v in vals], dtype=np.uint16)
    np.testing.assert_array_equal(got, want)
    halves = np.arange(0, 65536, 7, dtype=np.uint16)
    back = np.array([L.o_f16_to_f32(int(h)) for h in halves], dtype=np.float32)
    ref = halves.view(np.float16).astype(np.float32)
    np.testing.assert_array_equal(back[~np.isnan(ref)], ref[~np.isnan(ref)])
    assert np.isnan(back[np.isnan(ref)]).all()
