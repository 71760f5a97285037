"""T3 of tests/test_gpu_parity.py on the usual-glTF-set scene, per library build: python tools/gpu_debug_t3.py lib.so ..."""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, os.environ["TR_ROOT"])
import numpy as np, torch, ctypes as C
from transmission_renderer_amd import _lib
_lib.LIB_PATH = os.environ["TR_AB_LIB"]
class _Tolerant(C.CDLL):
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name.startswith("tr_"):
                return type("missing", (), {})()
            raise
C.CDLL = _Tolerant
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut
from oracle import oracle
w, h, nl = 256, 192, 2
scene = synthetic.make_scene(w, h, num_point_lights=nl, coverage="full", textured=True)
for m in scene["materials"]:
    t = m.textures
    t.emissive = t.transmission = t.thickness = t.specular = t.specular_colour = -1
lut = load_ggx_lut()
r = TransmissionRenderer(0); dev = r.device
r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"]); r.upload_ggx_lut(lut)
r.upload_textures(scene["textures"])
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev), torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
g = GBufferPlanes.from_numpy(scene["gbuffer"], dev)
b = oracle.SceneBinding(scene, lut)
tex = oracle.new_pyramid(w, h, synthetic.make_opaque_mip0(w, h)); oracle.generate_mips(w, h, tex)
pyr = OpaquePyramid(w, h, dev); pyr.texels.copy_(torch.from_numpy(tex).to(dev))
base = np.full((h, w, 4), 0.125, dtype=np.float32)
t32 = torch.from_numpy(base).to(dev)
r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, t32); torch.cuda.synchronize()
_, o32 = oracle.shade_transmission(b, scene["gbuffer"], tex, hdr_f16=base.astype(np.float16), hdr_f32=base.copy(), nthreads=8)
_, o64 = oracle.shade_transmission(b, scene["gbuffer"], tex, hdr_f16=base.astype(np.float16), hdr_f32=base.astype(np.float64), nthreads=8, fp64=True)
got = t32.cpu().numpy().astype(np.float64)
ne = lambda a, c: (a - c) / np.maximum(np.abs(c), 1.0)
noise = np.abs(ne(o32.astype(np.float64), o64)).max(axis=2)
off = np.abs(ne(got, o64)).max(axis=2)
bad = noise > 1e-5
viol = bad & (off > noise + 1e-4)
print(os.path.basename(os.environ["TR_AB_LIB"]), "ill", int(bad.sum()), "violations", int(viol.sum()), "max gpu off", off.max(), "rmse", np.sqrt((ne(got, o64) ** 2).mean()))
for (y, x) in np.argwhere(viol)[:6]:
    print("   ", y, x, "mat", scene["gbuffer"]["material_id"][y, x], "gpu", got[y, x, :3], "o32", o32[y, x, :3], "o64", o64[y, x, :3], "noise", noise[y, x], "off", off[y, x])
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, TR_ROOT=ROOT, TR_AB_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(out.stdout[-3000:], out.stderr[-1500:] if out.returncode else "")
