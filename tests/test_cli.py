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


def test_cli_resolves_bare_names_like_the_reference(tmp_path, capsys):
    """path_for_gltf_model (src/model_loading.rs:381-390): <dir>/2.0/<Name>/glTF/<Name>.gltf; --external-model takes a path as it is."""
    assert cli.main(["Nothing", "--sample-models-dir", str(tmp_path)]) == 2
    assert os.path.join(str(tmp_path), "2.0", "Nothing", "glTF", "Nothing.gltf") in capsys.readouterr().err
    assert cli.main([str(tmp_path / "model.bin"), "--external-model"]) == 2      # a path by declaration: "no such file"
    assert "no such file" in capsys.readouterr().err
    assert cli.main(["meshes", "--frames", "0"]) == 2


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


@pytest.mark.gpu
def test_cli_frame_loop_rotates_the_model_and_the_spotlights(tmp_path):
    """--rotate-model / --spotlights with --frames N: the reference's per-frame rewrites (src/main.rs:1243-1261, 1316-1322) through
    tr_update_instances / tr_update_lights — the last frame differs from the first, a loop without the flags does not; and a
    model behind --external-model (a path without a glTF suffix) loads like the same file with one."""
    import shutil
    import make_demo_gltf
    from transmission_renderer_amd.png import read_png_rgba8
    out = [str(tmp_path / f"{k}.png") for k in range(6)]
    common = ["--width", "320", "--height", "180", "--spotlights"]
    assert cli.main(["meshes", *common, "--out", out[0]]) == 0
    assert cli.main(["meshes", *common, "--frames", "40", "--out", out[1]]) == 0
    assert (read_png_rgba8(out[0]) != read_png_rgba8(out[1])).any(), "the spotlights did not turn"          # (0.4 rad after 40 frames)
    model = str(tmp_path / "demo.glb")
    make_demo_gltf.main(model)
    renamed = str(tmp_path / "model.bin")
    shutil.copy(model, renamed)
    view = ["--width", "320", "--height", "180", "--scale", "0.15"]
    assert cli.main([model, *view, "--out", out[2]]) == 0
    assert cli.main([renamed, "--external-model", *view, "--out", out[3]]) == 0
    assert (read_png_rgba8(out[2]) == read_png_rgba8(out[3])).all()
    # the LAST instance of the model buffers gets Quat::from_rotation_y(-0.0025 k) as its rotation (src/main.rs:920-921, 1258-1261)
    assert cli.main([model, *view, "--frames", "40", "--out", out[4]]) == 0
    assert cli.main([model, *view, "--frames", "40", "--rotate-model", "--out", out[5]]) == 0
    assert (read_png_rgba8(out[2]) == read_png_rgba8(out[4])).all()             # a loop without the flags repeats the frame
    assert (read_png_rgba8(out[4]) != read_png_rgba8(out[5])).any(), "the model did not rotate"
