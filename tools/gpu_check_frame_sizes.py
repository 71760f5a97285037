"""One-off check on the GPU box: tr_record_frame (visibility-word shading) against the stepwise sequence through the
TGB-v1 planes, bit for bit, at 4K and 8K on the demo glTF.   python tools/gpu_check_frame_sizes.py demo.glb"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import gltf, meshes, synthetic, wire
from transmission_renderer_amd.renderer import OpaquePyramid, TransmissionRenderer

loaded = gltf.load_gltf(sys.argv[1], base_transform=meshes.Similarity(np.array([0.0, 2.0, 0.0], np.float32), 1.0))
geo = loaded.geometry()
r = TransmissionRenderer(0)
r.upload_ggx_lut()
for (w, h) in ((3840, 2160), (7680, 4320)):
    view = wire.default_camera()[1]
    sc = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
    sc["materials"], sc["textures"] = loaded.materials, loaded.textures
    q = wire.view_rotation_inverse(view)
    culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    r.upload_materials(sc["materials"]); r.upload_textures(sc["textures"]); r.upload_lights(sc["lights"]); r.upload_geometry(geo)
    aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
    work = r.new_frame_buffers(w, h)
    for _ in range(2):
        hdr, ldr = r.record_frame(sc["uniforms"], sc["push"], culling, view, q, aabbs, work)
        torch.cuda.synchronize()
    got = hdr.clone()
    del work
    r.assign_lights_to_clusters(view, q, aabbs)
    o, t = r.new_layer(w, h), r.new_layer(w, h)
    r.draw_scene(culling, sc["push"], o, t)
    pyr = OpaquePyramid(w, h, r.device)
    want = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
    r.record(o, t, sc["uniforms"], sc["push"], want, pyr)
    torch.cuda.synchronize()
    same = torch.equal(got.view(torch.int16), want.view(torch.int16))
    cov = (got[..., :3].float().sum(dim=2) > 0).float().mean().item()
    print(f"{w}x{h}: frame recorder == stepwise sequence: {same}; covered {cov:.2f}")
    del o, t, pyr, want, got
    torch.cuda.empty_cache()
