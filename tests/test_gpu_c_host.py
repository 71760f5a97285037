"""A host that is NOT Python (run with -m gpu): examples/host_frame.c — plain C11, gcc, linked against libtr_shade.so and the
HIP runtime only — renders three frames of the procedural mesh scene through the C ABI (uploads, tr_write_cluster_data,
per-frame tr_update_instances / tr_update_lights, tr_record_frame) in its own process, without torch having loaded the HIP
runtime first.  Its presented frame and HDR target must be, byte for byte, what the Python path produces for the same
scene and the same per-frame rewrites."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from transmission_renderer_amd import _lib, meshes, synthetic, wire  # noqa: E402
from test_gpu_raster import _scene  # noqa: E402
from test_gpu_updates import _rotated, _spot  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "host_frame")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "host_frame.c"), "-L" + os.path.dirname(_lib.LIB_PATH), "-ltr_shade", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    needed = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libtr_shade.so" in needed and "python" not in needed.lower() and "torch" not in needed.lower(), needed
    return exe


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).tobytes()


def _pad4(b):
    return b + b"\0" * (-len(b) % 4)


@pytest.mark.timeout(600)
def test_c_host_renders_the_python_paths_frames(tmp_path, ggx_lut):
    from transmission_renderer_amd.renderer import TransmissionRenderer
    w, h, frames = 640, 360, 3
    view = wire.default_camera()[1]
    geo = meshes.make_mesh_scene()
    sc = _scene(w, h, view)
    lights = synthetic.make_lights(2) + [_spot((0.5, 2.5, -2.5), (0.0, -1.0, 0.2), 0.6), _spot((-1.0, 2.0, -3.5), (0.3, -1.0, 0.0), 0.5)]
    q = wire.view_rotation_inverse(view)
    culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    inv_p = wire.inverse_perspective(w, h)
    lottes = wire.LottesParams()
    assert _lib.load().tr_lottes_defaults(C.byref(lottes)) == 0
    lut = np.ascontiguousarray(ggx_lut, dtype=np.uint8)
    first_inst, n_inst = 1, 2

    def rewrites(k):
        angle = 0.4 * k
        new_lights = [_spot((0.5, 2.5, -2.5), (np.sin(angle), -1.0, np.cos(angle)), 0.6),
                      _spot((-1.0, 2.0, -3.5), (np.sin(angle + np.pi), -1.0, np.cos(angle + np.pi)), 0.5)]
        return _rotated(geo["instances"], first_inst, n_inst, angle)[first_inst:first_inst + n_inst], new_lights

    inst = np.ascontiguousarray(geo["instances"], dtype=wire.INSTANCE_DTYPE)
    prim = np.ascontiguousarray(geo["primitives"], dtype=wire.PRIMITIVE_DTYPE)
    head = np.array([0x43535254, w, h, len(sc["materials"]), len(lights), len(geo["position"]), len(geo["index"]), len(prim), len(inst),
                     len(sc["textures"]), lut.shape[1], lut.shape[0]], dtype=np.uint32)
    blob = [head.tobytes(), bytes(sc["push"]), bytes(sc["uniforms"]), bytes(culling), _f32(np.asarray(view).reshape(-1)), _f32(q),
            _f32(np.asarray(inv_p).reshape(-1)), bytes(lottes),
            b"".join(bytes(m) for m in sc["materials"]), b"".join(bytes(l) for l in lights),
            _f32(geo["position"]), _f32(geo["normal"]), _f32(geo["uv"]), np.ascontiguousarray(geo["index"], dtype=np.uint32).tobytes(),
            prim.tobytes(), inst.tobytes()]
    for img, srgb in sc["textures"]:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        blob += [np.array([img.shape[1], img.shape[0], 1 if srgb else 0, 0], dtype=np.uint32).tobytes(), img.tobytes()]
    blob.append(lut.tobytes())
    for k in range(1, frames):
        ri, rl = rewrites(k)
        blob += [np.array([first_inst, n_inst], dtype=np.uint32).tobytes(), np.ascontiguousarray(ri).tobytes(),
                 np.array([2, 2], dtype=np.uint32).tobytes(), b"".join(bytes(l) for l in rl)]
    scene_path, out_path = tmp_path / "scene.bin", tmp_path / "frame.rgba8"
    with open(scene_path, "wb") as f:
        f.write(b"".join(_pad4(b) for b in blob))

    exe = _build(tmp_path)
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    p = subprocess.run([exe, str(scene_path), str(out_path), str(frames)], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, (p.stdout, p.stderr)
    assert "host_frame: 3 frame(s) of 640x360" in p.stdout
    got_ldr = np.fromfile(out_path, dtype=np.uint8).reshape(h, w, 4)
    got_hdr = np.fromfile(str(out_path) + ".hdr16", dtype=np.uint16).reshape(h, w, 4)

    # the Python path, same scene, same rewrites
    r = TransmissionRenderer(0)
    try:
        r.upload_ggx_lut(lut)
        r.upload_materials(sc["materials"])
        r.upload_textures(sc["textures"])
        r.upload_lights(lights)
        r.upload_geometry(geo)
        aabbs = r.write_cluster_data(sc["uniforms"], inv_p, (w, h))
        work = r.new_frame_buffers(w, h)
        for k in range(frames):
            if k >= 1:
                ri, rl = rewrites(k)
                r.update_instances(first_inst, ri)
                r.update_lights(2, rl)
            hdr, ldr = r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work, lottes=lottes)
        torch.cuda.synchronize()
        want_ldr, want_hdr = ldr.cpu().numpy(), hdr.cpu().numpy().view(np.uint16)
    finally:
        r.close()
    assert (want_hdr[..., :3].view(np.float16).astype(np.float32).sum(axis=2) > 0).mean() > 0.2
    np.testing.assert_array_equal(got_hdr, want_hdr)
    np.testing.assert_array_equal(got_ldr, want_ldr)
