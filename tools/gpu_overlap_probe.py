#!/usr/bin/env python3
"""How much of the 4K transmissive launch is head / tail / inter-launch gap: the same K launches issued (a) back to back
on one stream, (b) alternately on S streams with S frame buffers — sharing ONE set of inputs (what round 3's first bench
did: the second frame finds the first one's plane reads in the caches, a frame of a real renderer does not) and with its
own copy of the G-buffer planes and the pyramid per frame in flight —, (c) every frame as two / four row bands on two /
four streams (disjoint parts of one frame: nothing shared).  Prints wall-clock us per frame for each.  python tools/gpu_overlap_probe.py [lights]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transmission_renderer_amd import synthetic  # noqa: E402
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer  # noqa: E402

w, h = 3840, 2160
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
all_t = len(sys.argv) > 2 and sys.argv[2] == "all"
r = TransmissionRenderer(0)
scene = synthetic.make_scene(w, h, num_point_lights=nl)
if all_t:
    for m in scene["materials"]:
        m.transmission_factor = 1.0
r.upload_ggx_lut()
r.upload_materials(scene["materials"])
r.upload_lights(scene["lights"])
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(r.device),
                     torch.from_numpy(scene["light_indices"].view(np.int32)).to(r.device))
g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
pyr = OpaquePyramid(w, h, r.device)
pyr.level(0).copy_(torch.from_numpy(synthetic.make_opaque_mip0(w, h)).to(r.device))
r.generate_mips(pyr)
S_MAX = 4
gs = [g] + [GBufferPlanes(g.pos_depth.clone(), g.nrm_scale.clone(), g.uv.clone(), g.material_id.clone()) for _ in range(S_MAX - 1)]
pyrs = [pyr]
for _ in range(S_MAX - 1):
    q = OpaquePyramid(w, h, r.device)
    q.texels.copy_(pyr.texels)
    pyrs.append(q)
hdrs = [torch.zeros((h, w, 4), dtype=torch.float16, device=r.device) for _ in range(S_MAX)]
streams = [torch.cuda.Stream() for _ in range(S_MAX)]
u, p = scene["uniforms"], scene["push"]


def ramp():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(16):
            r.shade_transmission(g, u, p, pyr, hdrs[0])
        torch.cuda.synchronize()


def wall(fn, K=400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6


def one_stream(K):
    for _ in range(K):
        r.shade_transmission(g, u, p, pyr, hdrs[0])


def multi(S):
    def fn(K):
        for k in range(K):
            with torch.cuda.stream(streams[k % S]):
                r.shade_transmission(g, u, p, pyr, hdrs[k % S])
    return fn


def multi_own(S):
    def fn(K):
        for k in range(K):
            with torch.cuda.stream(streams[k % S]):
                r.shade_transmission(gs[k % S], u, p, pyrs[k % S], hdrs[k % S])
    return fn


def parts(n):
    rows = ((h // n + 3) // 4) * 4
    def fn(K):
        for k in range(K):
            for i in range(n):
                with torch.cuda.stream(streams[i]):
                    r.shade_transmission(g, u, p, pyr, hdrs[0], (0, i * rows, w, min(h, (i + 1) * rows)))
    return fn


def strips(n, rows):
    """the frame as n launches of rank-interleaved strips (tr_set_strips): launch i shades strips i, i + n, ... of `rows` rows"""
    def fn(K):
        for k in range(K):
            for i in range(n):
                r.set_strips(rows, n, i)
                with torch.cuda.stream(streams[i]):
                    r.shade_transmission(g, u, p, pyr, hdrs[0], (0, 0, w, h))
        r.set_strips(0, 1, 0)
    return fn


res = {}
ramp()
for rep in range(3):
    for name, fn in (("one_stream", one_stream), ("two_streams_shared_inputs", multi(2)), ("two_streams_own_inputs", multi_own(2)),
                     ("three_streams_own_inputs", multi_own(3)), ("halves_two_streams", parts(2)), ("quarters_four_streams", parts(4)),
                     ("strips136_two_streams", strips(2, 136)), ("strips68_two_streams", strips(2, 68)), ("strips272_two_streams", strips(2, 272))):
        fn(50)
        res.setdefault(name, []).append(round(wall(fn), 2))
print(json.dumps({"lights": nl, "all_transmissive": all_t, "us_per_frame": res}))
