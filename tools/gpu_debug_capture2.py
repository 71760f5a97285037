"""Debug: tr_frustum_culling (a memset + a kernel of atomics) captured into a HIP graph (run on the GPU box)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import meshes, synthetic, wire
from transmission_renderer_amd.renderer import TransmissionRenderer
r = TransmissionRenderer(0)
w, h = 256, 128
view = wire.default_camera()[1]
geo = meshes.make_mesh_scene()
culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
prim = torch.from_numpy(np.frombuffer(geo["primitives"].tobytes(), dtype=np.uint8).copy()).cuda()
inst = torch.from_numpy(np.frombuffer(geo["instances"].tobytes(), dtype=np.uint8).copy()).cuda()
n_prim = prim.numel() // wire.PRIMITIVE_DTYPE.itemsize
n_inst = inst.numel() // wire.INSTANCE_DTYPE.itemsize
counts = torch.zeros(n_prim, dtype=torch.int32, device="cuda")
def cull():
    r._check(r.lib.tr_frustum_culling(r._ctx, prim.data_ptr(), n_prim, inst.data_ptr(), n_inst, C.byref(culling), counts.data_ptr(), r._stream()), "cull")
cull(); torch.cuda.synchronize(); want = counts.clone(); print("direct", want.tolist())
side = torch.cuda.Stream()
with torch.cuda.stream(side): cull()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side): cull()
for k in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay", k, counts.tolist())
# the same with torch's own zero_ inside the capture
x = torch.zeros(8, dtype=torch.int32, device="cuda")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=side):
    x.zero_(); x.add_(1)
for k in range(3):
    g2.replay(); torch.cuda.synchronize(); print("torch zero+add replay", k, x.tolist())
