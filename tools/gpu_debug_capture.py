"""Debug: tr_draw_scene captured into a HIP graph vs direct (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from transmission_renderer_amd import meshes, synthetic, wire
from transmission_renderer_amd.renderer import TransmissionRenderer
r = TransmissionRenderer(0)
w, h = 256, 128
view = wire.default_camera()[1]
geo = meshes.make_mesh_scene()
sc = synthetic.make_scene(w, h, num_point_lights=1, textured=True, with_gbuffer=False)
culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
r.upload_ggx_lut(); r.upload_materials(sc["materials"]); r.upload_textures(sc["textures"]); r.upload_geometry(geo)
o, t = r.new_layer(w, h), r.new_layer(w, h)
r.draw_scene(culling, sc["push"], o, t); torch.cuda.synchronize()
want = o.material_id.clone(); wantt = t.material_id.clone()
for k in range(3):
    o.material_id.fill_(-7); r.draw_scene(culling, sc["push"], o, t); torch.cuda.synchronize()
    print("direct", k, int((o.material_id != want).sum()))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    r.draw_scene(culling, sc["push"], o, t)
torch.cuda.synchronize()
print("side", int((o.material_id != want).sum()))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    r.draw_scene(culling, sc["push"], o, t)
for k in range(4):
    o.material_id.fill_(-7); t.material_id.fill_(-7)
    g.replay(); torch.cuda.synchronize()
    d = o.material_id != want
    print("replay", k, int(d.sum()), int((t.material_id != wantt).sum()), "untouched(-7):", int((o.material_id == -7).sum()), "rows with diffs:", torch.nonzero(d.any(dim=1)).flatten()[:10].tolist())
vals, cnt = torch.unique(o.material_id, return_counts=True)
print("replayed ids:", vals.tolist(), cnt.tolist())
vals, cnt = torch.unique(want, return_counts=True)
print("wanted ids:", vals.tolist(), cnt.tolist())
r.draw_scene(culling, sc["push"], o, t); torch.cuda.synchronize()
print("direct after replays", int((o.material_id != want).sum()))
g.replay(); torch.cuda.synchronize(); print("replay after direct", int((o.material_id != want).sum()))
g.replay(); torch.cuda.synchronize(); print("replay again", int((o.material_id != want).sum()))
