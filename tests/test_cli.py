"""The glTF-in/frame-out command line (the reference's `Opt`, src/main.rs:65-91, plus --backdrop for the scene it always
loads behind the model, :342-351).  Argument handling on CPU; a rendered frame with and without a backdrop on the GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from transmission_renderer_amd import cli, gltf  # noqa: E402


def test_cli_refuses_bad_scene_arguments(tmp_path, capsys):
    assert cli.main(["DragonAttenuation"]) == 2                      # bare sample-model names need the Khronos checkout
    assert cli.main([str(tmp_path / "missing.glb")]) == 2
    import make_demo_gltf
    model = str(tmp_path / "model.glb")
    make_demo_gltf.main(model)
    assert cli.main([model, "--backdrop", str(tmp_path / "nothing.glb")]) == 2
    assert cli.main(["meshes", "--backdrop", model]) == 2            # a backdrop goes behind a glTF model
    assert "backdrop" in capsys.readouterr().err


def test_backdrop_scene_is_loaded_first_and_model_appended(tmp_path):
    """What --backdrop does on the host: the reference's two load_gltf calls into one set of model buffers."""
    import make_demo_gltf
    from transmission_renderer_amd import meshes
    path = str(tmp_path / "demo.glb")
    make_demo_gltf.main(path)
    back = gltf.load_gltf(path)
    n_mat, n_prim, n_tex = len(back.materials), len(back.geometry()["primitives"]), len(back.textures)
    both = gltf.load_gltf(path, scene=back, base_transform=meshes.Similarity(np.array([0.0, 2.0, 0.0], np.float32), 0.5),
                          roughness_override=0.25)
    g = both.geometry()
    assert len(both.materials) == 2 * n_mat and len(g["primitives"]) == 2 * n_prim
    assert len(both.textures) == 2 * n_tex                            # each load uploads its own images (per-call cache)
    assert both.materials[0].roughness_factor != 0.25 and both.materials[n_mat].roughness_factor == 0.25
    assert g["instances"]["material_id"].max() >= n_mat               # the model's instances point at the appended materials


@pytest.mark.gpu
def test_cli_renders_a_model_in_front_of_a_backdrop(tmp_path):
    import make_demo_gltf
    from transmission_renderer_amd.png import read_png_rgba8
    model, out_a, out_b = str(tmp_path / "demo.glb"), str(tmp_path / "a.png"), str(tmp_path / "b.png")
    make_demo_gltf.main(model)
    assert cli.main([model, "--width", "320", "--height", "180", "--scale", "0.15", "--out", out_a]) == 0
    assert cli.main([model, "--backdrop", model, "--width", "320", "--height", "180", "--scale", "0.15", "--out", out_b]) == 0
    a, b = read_png_rgba8(out_a), read_png_rgba8(out_b)
    assert a.shape == b.shape == (180, 320, 4)
    changed = (a[..., :3] != b[..., :3]).any(axis=2).mean()
    assert changed > 0.1, changed            # the backdrop fills pixels the model alone leaves to the clear colour


@pytest.mark.gpu
def test_cli_tonemap_constants_are_explicit_inputs(tmp_path):
    """The operator's constants are command-line inputs (the reference's come from an un-vendored crate's Default impl):
    the defaults reproduce the frame without flags, another contrast changes it, and the presented frame equals the
    oracle's fragment_tonemap of the HDR frame under the same constants."""
    import ctypes as C
    from oracle import oracle
    from transmission_renderer_amd import _lib, wire
    from transmission_renderer_amd.png import read_png_rgba8
    a, b, c = (str(tmp_path / n) for n in ("a.png", "b.png", "c.png"))
    hdr = str(tmp_path / "c.npy")
    base = ["synthetic", "--width", "160", "--height", "96"]
    assert cli.main(base + ["--out", a]) == 0
    assert cli.main(base + ["--out", b, "--tonemap-contrast", "1.6", "--tonemap-shoulder", "0.977", "--tonemap-hdr-max", "8",
                            "--tonemap-mid-in", "0.18", "--tonemap-mid-out", "0.267", "--tonemap-crosstalk", "4"]) == 0
    assert cli.main(base + ["--out", c, "--hdr-out", hdr, "--tonemap-contrast", "1.2", "--tonemap-hdr-max", "16"]) == 0
    fa, fb, fc = read_png_rgba8(a), read_png_rgba8(b), read_png_rgba8(c)
    assert (fa == fb).all()
    assert (fa[..., :3] != fc[..., :3]).any(axis=2).mean() > 0.5
    q, tm = wire.LottesParams(), wire.TonemapParams()
    lib = _lib.load()
    lib.tr_lottes_defaults(C.byref(q))
    q.contrast, q.hdr_max = 1.2, 16.0
    q.saturation, q.cross_saturation = q.contrast, q.contrast * 16.0
    lib.tr_bake_lottes_params(C.byref(q), C.byref(tm))
    want = oracle.tonemap_frame(np.ascontiguousarray(np.load(hdr), dtype=np.float16), tm)[0]
    assert np.abs(fc[..., :3].astype(np.int32) - want[..., :3].astype(np.int32)).max() <= 1
