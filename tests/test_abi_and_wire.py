"""The C-ABI library loads and exports exactly what include/tr_shade.h declares; the ctypes twins and the C
header agree with the reference's wire layouts.  No GPU needed: nothing here computes."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from transmission_renderer_amd import _lib, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tr_shade.h")


def _ensure_built():
    if not os.path.exists(_lib.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()


def test_header_declares_what_the_library_exports():
    _ensure_built()
    text = open(HEADER).read()
    declared = set(re.findall(r"^(?:tr_status|const char\*|int32_t|uint32_t)\s+(tr_[a-z0-9_]+)\s*\(", text, re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = C.CDLL(_lib.LIB_PATH)   # dlopen only; libamdhip64 is present without a GPU
    for sym in _lib.SYMBOLS:
        assert getattr(lib, sym) is not None
    lib.tr_abi_version.restype = C.c_uint32
    assert lib.tr_abi_version() == 1
    lib.tr_status_string.restype = C.c_char_p
    assert b"no HIP device" in lib.tr_status_string(2)


def test_product_library_does_not_link_the_oracle():
    _ensure_built()
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert " o_" not in syms  # no oracle entry point is compiled into the product


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the context cannot be created (status TR_ERR_NO_DEVICE): there is no CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    ctx = C.c_void_p()
    st = lib.tr_context_create(0, C.byref(ctx))
    assert st == 2 and not ctx.value
    from transmission_renderer_amd.renderer import TransmissionRenderer
    with pytest.raises(RuntimeError):
        TransmissionRenderer(0)


def test_header_static_asserts_compile_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "tr_shade.h"\nint main(void){return 0;}\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "t2.o")])


def test_ctypes_layouts_match_reference_offsets():
    # shared-structs/src/lib.rs (offsets as decorated in compiled-shaders/normal/fragment_transmission.spv)
    assert C.sizeof(wire.PushConstants) == 96
    assert (wire.PushConstants.view_position.offset, wire.PushConstants.framebuffer_size.offset,
            wire.PushConstants.acceleration_structure_address.offset) == (64, 80, 88)
    assert C.sizeof(wire.Uniforms) == 96
    u = wire.Uniforms
    assert (u.sun_dir.offset, u.sun_intensity.offset, u.cluster_size_in_pixels.offset, u.num_clusters.offset,
            u.debug_clusters.offset, u.ggx_lut_texture_index.offset) == (32, 48, 64, 72, 80, 84)
    m = wire.MaterialInfo
    assert C.sizeof(m) == 160
    assert (m.metallic_factor.offset, m.roughness_factor.offset, m.alpha_clipping_cutoff.offset,
            m.diffuse_factor.offset, m.emissive_factor.offset, m.normal_map_scale.offset,
            m.occlusion_strength.offset, m.index_of_refraction.offset, m.transmission_factor.offset,
            m.thickness_factor.offset, m.attenuation_distance.offset, m.attenuation_colour.offset,
            m.specular_factor.offset, m.specular_colour_factor.offset) == (36, 40, 44, 48, 64, 80, 84, 88, 92, 96,
                                                                            100, 112, 128, 144)
    assert C.sizeof(wire.Light) == 48 and C.sizeof(wire.ClusterAabb) == 32
    lib = _lib.load() if os.path.exists(_lib.LIB_PATH) else None
    if lib is not None:  # tr_pyramid_layout is host-only arithmetic
        p, n = wire.Pyramid(), C.c_size_t()
        assert lib.tr_pyramid_layout(3840, 2160, C.byref(p), C.byref(n)) == 0
        levels, layout, total = wire.pyramid_layout(3840, 2160)
        assert p.levels == levels == 12 and n.value == (total + 1) * 8   # + one texel of tail padding
        assert [p.level_offset[l] for l in range(levels)] == [o for o, _, _ in layout]
        assert layout[-1][1:] == (1, 1) and layout[5][1:] == (120, 67)


def test_material_defaults_and_light_constructors():
    m = wire.MaterialInfo.default()
    assert m.index_of_refraction == 1.5 and m.transmission_factor == 0.0 and m.attenuation_distance == float("inf")
    assert list(m.attenuation_colour) == [1, 1, 1] and m.specular_factor == 1.0 and m.textures.diffuse == -1
    l = wire.Light.new_point((0.0, 0.8, 0.0), (1, 0, 0), 5.0)
    assert list(l.colour_emission_and_falloff_distance_sq) == [5.0, 0.0, 0.0, pytest.approx(100.0)]
    assert l.spotlight_direction_and_outer_angle[3] == 0.0   # point light
    s = wire.Light.new_spot((0, 4, 0), (1, 1, 0.5), 50.0, (0, 0, 1), 0.7, 0.8)
    assert s.position_and_spotlight_epsilon[3] == pytest.approx(np.cos(np.float32(0.7)) - np.cos(np.float32(0.8)), rel=1e-6)
    assert s.spotlight_direction_and_outer_angle[3] == pytest.approx(0.8)


def test_push_constants_project_the_gbuffer_back_onto_its_pixels():
    """proj_view (reversed-Z, y-flipped, src/main.rs:39-54) maps the synthetic world positions to their pixels."""
    from transmission_renderer_amd import synthetic
    w, h = 96, 64
    g = synthetic.make_gbuffer(w, h)
    pc = wire.make_push_constants(w, h)
    pv = np.array(list(pc.proj_view), dtype=np.float64).reshape(4, 4)  # [column][row]
    pos = g["pos_depth"][..., :3].astype(np.float64)
    clip = pos @ pv[:3, :] + pv[3, :]
    ndc = clip[..., :2] / clip[..., 3:4]
    xs = (ndc[..., 0] + 1) / 2 * w
    ys = (ndc[..., 1] + 1) / 2 * h
    np.testing.assert_allclose(xs, np.broadcast_to(np.arange(w)[None, :] + 0.5, xs.shape), atol=2e-3)
    np.testing.assert_allclose(ys, np.broadcast_to(np.arange(h)[:, None] + 0.5, ys.shape), atol=2e-3)
    np.testing.assert_allclose(clip[..., 2] / clip[..., 3], g["pos_depth"][..., 3], rtol=2e-4)


def test_the_c_host_example_compiles_and_links_without_python(tmp_path):
    """examples/host_frame.c — the boundary driven from plain C11 — builds with gcc -Wall -Werror against include/tr_shade.h and
    links against libtr_shade.so and the HIP runtime only (tests/test_gpu_c_host.py runs it on the GPU box)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("no gcc / HIP runtime to link against")
    from transmission_renderer_amd import _lib
    exe = str(tmp_path / "host_frame")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", "host_frame.c"), "-L" + os.path.dirname(_lib.LIB_PATH), "-ltr_shade", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    needed = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libtr_shade.so" in needed and "python" not in needed.lower() and "torch" not in needed.lower(), needed
