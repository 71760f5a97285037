"""Whole-frame throughput with S frames in flight (run on the GPU box): S contexts (each its own work buffers: visibility
words, triangle records, pyramid, targets) on S HIP streams, frame k recorded into context k mod S — what a renderer with a
swapchain does.  One frame's launches are a dependent chain of ~13 kernels, five of them latency-bound (~4.5 us each) and
every one with a straggler tail; a second frame fills those gaps.
    python tools/gpu_bench_frames_in_flight.py [scene.glb | meshes] [S ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import gltf, meshes, synthetic, wire
from transmission_renderer_amd.renderer import TransmissionRenderer

name = sys.argv[1] if len(sys.argv) > 1 else "meshes"
counts = [int(a) for a in sys.argv[2:]] or [1, 2, 3]
w, h = 3840, 2160


def make_context():
    r = TransmissionRenderer(0)
    scene = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False, textured=(name == "meshes"))
    if name == "meshes":
        geometry = meshes.make_mesh_scene(extra_instances=True)
        scene["materials"][2].alpha_clipping_cutoff = 0.75
        scene["materials"][7].alpha_clipping_cutoff = 0.6
    else:
        loaded = gltf.load_gltf(name, base_transform=meshes.Similarity(np.array([0.0, 2.0, 0.0], np.float32), 1.0))
        geometry = loaded.geometry()
        scene["materials"] = loaded.materials or [wire.MaterialInfo.default()]
        scene["textures"] = loaded.textures
    r.upload_ggx_lut(); r.upload_materials(scene["materials"])
    if scene.get("textures"): r.upload_textures(scene["textures"])
    r.upload_lights(scene["lights"]); r.upload_geometry(geometry)
    _, view = wire.default_camera()
    aabbs = r.write_cluster_data(scene["uniforms"], wire.inverse_perspective(w, h), (w, h))
    culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    work = r.new_frame_buffers(w, h)
    q = wire.view_rotation_inverse(view)
    return r, (lambda: r.record_frame(scene["uniforms"], scene["push"], culling, view, q, aabbs, work))


ctxs = [make_context() for _ in range(max(counts))]
torch.cuda.synchronize()
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(max(counts) - 1)]
for S in counts:
    def run(n):
        for k in range(n):
            with torch.cuda.stream(streams[k % S]):
                ctxs[k % S][1]()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        run(8); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(60)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t1) / 60)
    ts.sort()
    print(f"{name} {w}x{h}: {S} frame(s) in flight: {ts[len(ts) // 2] * 1e6:.1f} us per frame ({1 / ts[len(ts) // 2]:.0f} frames/s)")
