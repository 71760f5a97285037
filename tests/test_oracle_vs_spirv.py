"""Pins the CPU oracle against the reference's OWN compiled shaders.

tests/golden/spirv_case_*.npz hold the outputs of compiled-shaders/normal/fragment_transmission.spv and
fragment.spv (the build output the reference commits) executed instruction by instruction by
oracle/spirv_ref/spirv_interp.py on seeded inputs (generator: tools/make_golden_spirv.py).  The fixed-function
texel filtering the SPIR-V delegates to Vulkan was answered by the oracle's own sampling functions, so these
fixtures pin every arithmetic operation, its order, the cluster light loop and the descriptor/push-constant byte
layouts of oracle/tr_oracle.c — bit for bit."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from oracle import oracle
from transmission_renderer_amd import wire

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spirv_case_*.npz")))


def _scene_from_fixture(z):
    n_mat = z["materials"].size // C.sizeof(wire.MaterialInfo)
    n_light = z["lights"].size // C.sizeof(wire.Light)
    mats = [wire.MaterialInfo.from_buffer_copy(z["materials"].tobytes(), i * 160) for i in range(n_mat)]
    lights = [wire.Light.from_buffer_copy(z["lights"].tobytes(), i * 48) for i in range(n_light)]
    counts = z["cluster_counts"].astype(np.uint32)
    idx = np.tile(z["light_list"].astype(np.uint32)[None, :], (counts.size, 1)).reshape(-1)
    scene = {"materials": mats, "lights": lights, "cluster_counts": counts, "light_indices": idx,
             "uniforms": wire.Uniforms.from_buffer_copy(z["uniforms"].tobytes()),
             "push": wire.PushConstants.from_buffer_copy(z["push"].tobytes())}
    if len(z["texture_srgb"]):
        scene["textures"] = [(z[f"texture_{i}"], bool(sr)) for i, sr in enumerate(z["texture_srgb"])]
    w, h = int(z["width"]), int(z["height"])
    g = {"pos_depth": np.ascontiguousarray(z["pos_depth"]), "nrm_scale": np.ascontiguousarray(z["nrm_scale"]),
         "uv": np.ascontiguousarray(z["uv"]), "material_id": np.ascontiguousarray(z["material_id"]),
         "width": w, "height": h}
    return scene, g, w, h


def _ulps(a, b):
    a = a.astype(np.float32).view(np.int32).astype(np.int64)
    b = b.astype(np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_reference_spirv(path, ggx_lut):
    assert GOLDEN, "fixtures missing"
    z = np.load(path)
    scene, g, w, h = _scene_from_fixture(z)
    b = oracle.SceneBinding(scene, ggx_lut)
    tex = oracle.new_pyramid(w, h, z["opaque_mip0"])
    oracle.generate_mips(w, h, tex)
    _, t32 = oracle.shade_transmission(b, g, tex)
    _, o32, _ = oracle.shade_opaque(b, g)
    py, px = z["pixels"][:, 0], z["pixels"][:, 1]
    for name, got, want in (("fragment_transmission", t32[py, px], z["spirv_fragment_transmission"]),
                            ("fragment.hdr", o32[py, px], z["spirv_fragment_hdr"]),
                            ("fragment.opaque_sampled", o32[py, px], z["spirv_fragment_opaque_sampled"])):
        assert np.isfinite(want).all() and want.shape == got.shape
        u = _ulps(got, want)
        assert u.max() == 0, (name, "max ulp", int(u.max()), "mismatching values", int((u != 0).sum()), "of", u.size)


def test_fixture_covers_the_path():
    """The fixtures exercise what they claim: several materials, lights in and out of clusters, spotlights."""
    zs = [np.load(p) for p in GOLDEN]
    assert len(zs) >= 2
    assert all(len(np.unique(z["material_id"])) >= 8 for z in zs)
    assert any(len(np.unique(z["cluster_counts"])) > 1 for z in zs)             # lists of different length
    assert any((np.frombuffer(z["lights"].tobytes(), dtype=np.float32).reshape(-1, 12)[:, 11] != 0).any() for z in zs)
    # one case runs the material-texture path: implicit-LOD fetches, sRGB decode, normal mapping (OpDPdx/OpDPdy)
    tz = [z for z in zs if len(z["texture_srgb"])]
    assert tz
    mats = np.frombuffer(tz[0]["materials"].tobytes(), dtype=np.int32).reshape(-1, 40)[:, :9]
    assert (mats[:, 2] >= 0).any() and (mats[:, 0] >= 0).any() and (mats != -1).sum() >= 12
