"""The frame geometry bench.py uses at --gpus 8 (7680x8640, row bands of 1080 rows), first and last band, on ONE GPU:\nno limit is hit, the band is finite and opaque, the rows of other ranks stay untouched.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(sys.path[0], "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from transmission_renderer_amd import synthetic
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut
world = 8
fw, fh = bench.frame_size_for(world, 3840, 2160)
r = TransmissionRenderer(0); dev = r.device
scene = synthetic.make_scene(fw, fh, num_point_lights=1, with_gbuffer=False)
lut = load_ggx_lut(); r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"]); r.upload_ggx_lut(lut)
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev), torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
pyr = OpaquePyramid(fw, fh, dev); pyr.level(0).copy_(bench.make_mip0_torch(fw, fh, dev)); r.generate_mips(pyr)
hdr = torch.zeros((fh, fw, 4), dtype=torch.float16, device=dev)
for rank in (0, 7):
    band_rows = fh // world; y0, y1 = rank * band_rows, (rank + 1) * band_rows
    band = synthetic.make_gbuffer(fw, fh, rows=(y0, y1)); g = GBufferPlanes.from_numpy(band, dev)
    r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr, (0, y0, fw, y1)); torch.cuda.synchronize()
    got = hdr[y0:y1].float().cpu().numpy()
    assert np.isfinite(got).all() and (got[..., 3] == 1).all(), rank
    print("rank", rank, "frame", fw, fh, "band", y0, y1, "mean", float(got[..., :3].mean()), "ok")
print("untouched rows stay zero:", float(hdr[band_rows:7 * band_rows].abs().max()))
