"""The N > 1 path on the ONE GPU of the test box (run with -m gpu): two ranks, both on cuda:0, device tensors, real
TransmissionRenderer contexts.  RCCL refuses two ranks on one device, gloo does not: the process group is gloo and the
exchanges are staged through host memory (sharded.Compositor.host_staged) — everything else is the multi-GPU code as it runs
on eight GPUs: band origins of the plane sets, padded gather buffers, tr_generate_mips_band, tap windows on device pyramids,
the halo's fallback and its late verdict, rank-interleaved strips.  Criterion: every rank ends with the single-rank frame, bit
for bit.  (The CPU twin of this file, tests/test_sharding_cpu.py, drives the same recorder with an oracle-backed renderer.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["TR_ROOT"])
import numpy as np
import torch
import torch.distributed as dist
import bench
from transmission_renderer_amd import sharded, synthetic
from transmission_renderer_amd.renderer import OpaquePyramid, TransmissionRenderer, load_ggx_lut

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
report = {}


def context(w, h, thickness_scale=1.0):
    r = TransmissionRenderer(0)
    scene = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
    for m in scene["materials"]:
        m.thickness_factor *= thickness_scale
    r.upload_materials(scene["materials"])
    r.upload_lights(scene["lights"])
    r.upload_ggx_lut(load_ggx_lut())
    r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev),
                         torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
    return r, scene


def single_rank_frame(r, scene, w, h):
    g = bench.make_gbuffer_torch(w, h, dev)
    pyr = OpaquePyramid(w, h, dev)
    hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=dev)
    r.record(g, g, scene["uniforms"], scene["push"], hdr, pyr)
    torch.cuda.synchronize()
    return hdr, pyr


def same(a, b):
    return bool(torch.equal(a.contiguous().view(torch.int16), b.contiguous().view(torch.int16)))


# ---- A. the metric's step (BASELINE config 4): the transmissive pass in row bands + the composite, 1920x1080
w, h = 1920, 1080
r, scene = context(w, h)
comp = sharded.Compositor(world, rank, renderer=r)
assert comp.host_staged and comp.backend.startswith("torch.distributed:gloo"), comp.backend
rows, y0, y1 = sharded.band_rows(h, world, rank)
pyr = OpaquePyramid(w, h, dev)
pyr.level(0).copy_(bench.make_mip0_torch(w, h, dev))
r.generate_mips(pyr)
g_band = bench.make_gbuffer_torch(w, h, dev, rows=(y0, y1))
frame = torch.zeros((rows * world, w, 4), dtype=torch.float16, device=dev)
r.shade_transmission(g_band, scene["uniforms"], scene["push"], pyr, frame, (0, y0, w, y1))
ldr = torch.zeros((rows * world, w, 3), dtype=torch.uint8, device=dev)
tm = r.baked_tonemap_params()
r.tonemap_rgb8(frame[y0:y1], tm, out=ldr[y0:y1])
comp.allgather_rows(frame)
comp.allgather_rows(ldr)
whole = torch.zeros((h, w, 4), dtype=torch.float16, device=dev)
r.shade_transmission(bench.make_gbuffer_torch(w, h, dev), scene["uniforms"], scene["push"], pyr, whole)
whole_ldr = r.tonemap_rgb8(whole, tm)
torch.cuda.synchronize()
report["pass_1080p_hdr"] = same(frame[:h], whole)
report["pass_1080p_rgb8"] = bool(torch.equal(ldr[:h].reshape(-1), whole_ldr.reshape(-1)))
r.close()

# ---- B. the full pipeline (BASELINE config 5's band arithmetic at 1/8 of its size: 960x544, bands of 272 rows like 8K's at N = 8 ... of 2)
w, h = 960, 544
for name, kw, thickness_scale in (("gather", dict(exchange="allgather"), 1.0),
                                  ("halo", dict(exchange="halo"), 0.02),
                                  ("halo_fallback", dict(exchange="halo"), 1.0),
                                  ("halo_late", dict(exchange="halo", confirm="late"), 0.02)):
    r, scene = context(w, h, thickness_scale)
    want, _ = single_rank_frame(r, scene, w, h)
    comp = sharded.Compositor(world, rank, renderer=r)
    if "halo" in name:
        comp.halo_rows = 16 if name == "halo_fallback" else 96      # (thin volumes still throw taps tens of rows at this size)
    rows, y0, y1 = sharded.band_rows(h, world, rank)
    band = bench.make_gbuffer_torch(w, h, dev, rows=(y0, y1))
    frames = 3 if name == "halo_late" else 1
    for _ in range(frames):
        pyr = OpaquePyramid(w, h, dev, level0_rows=rows * world)
        pyr.texels.fill_(float("nan"))                      # (rows nobody delivers stay poison)
        hdr = torch.zeros((rows * world, w, 4), dtype=torch.float16, device=dev)
        sharded.record_sharded(r, band, band, scene["uniforms"], scene["push"], hdr, pyr, comp, **kw)
    if name == "halo_late":
        comp.confirm_halo()
    torch.cuda.synchronize()
    report[name] = same(hdr[:h], want)
    report[name + "_state"] = [comp.halo_fallbacks, comp.halo_inexact_frames, comp.halo_rows]
    r.close()

# ---- C. rank-interleaved strips of whole-frame buffers
r, scene = context(w, h)
want, _ = single_rank_frame(r, scene, w, h)
comp = sharded.Compositor(world, rank, renderer=r)
g = bench.make_gbuffer_torch(w, h, dev)
pyr = OpaquePyramid(w, h, dev)
hdr = torch.full((h, w, 4), -7.0, dtype=torch.float16, device=dev)
sharded.record_sharded_strips(r, g, g, scene["uniforms"], scene["push"], hdr, pyr, comp, strip_rows=64)
torch.cuda.synchronize()
report["strips"] = same(hdr, want)
r.close()

with open(os.path.join(os.environ["TR_OUT"], f"report_{rank}.json"), "w") as f:
    json.dump(report, f)
dist.barrier()
dist.destroy_process_group()
'''


def _spawn(code, tmp_path, world=2, timeout=900, extra_args=()):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TR_ROOT=ROOT, TR_OUT=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, "-c", code, *extra_args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout) + (p.returncode,))
        except subprocess.TimeoutExpired:
            for q in procs:          # the exact PIDs started above
                q.kill()
            raise
    return outs


@pytest.mark.timeout(1200)
def test_two_ranks_on_one_gpu_end_with_the_single_rank_frame(tmp_path):
    outs = _spawn(WORKER, tmp_path)
    for out, err, code in outs:
        assert code == 0, err[-4000:]
    for rank in range(2):
        rep = json.load(open(tmp_path / f"report_{rank}.json"))
        for key in ("pass_1080p_hdr", "pass_1080p_rgb8", "gather", "halo", "halo_fallback", "halo_late", "strips"):
            assert rep[key] is True, (rank, key, rep)
        assert rep["halo_state"][0] == 0, rep                         # thin volumes: the 96-row halo held
        assert rep["halo_fallback_state"][0] == 1 and rep["halo_fallback_state"][2] > 16, rep   # thick ones: redone, the halo grew
        assert rep["halo_late_state"][:2] == [0, 0], rep               # three frames confirmed a frame late, none inexact


@pytest.mark.timeout(1200)
def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 2 --one-device: run_rank with WORLD_SIZE = 2 on device tensors, its JSON line says which backend
    carried the composite; --require-rccl refuses the run (exit code 3) because that backend is not the library's RCCL."""
    port = _free_port()
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--steps", "4", "--warmup", "1", "--width", "1920",
            "--height", "1080", "--full-pipeline-size", "960x544", "--no-cpu-baseline"]

    def run(extra):
        procs = []
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen(base + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        return [p.communicate(timeout=900) + (p.returncode,) for p in procs]

    outs = run([])
    for out, err, code in outs:
        assert code == 0, err[-4000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["composite_backend"].startswith("torch.distributed:gloo") and "staged through host memory" in line["composite_backend"]
    assert line["rccl_ranks"] is None and "one_device_rehearsal" in line
    assert len(line["kernel_only"]["per_rank_kernel_ms"]) == 2
    fp = line["full_pipeline_8k"]
    assert fp["ms_per_frame"] > 0 and fp["exchange"] == "halo" and fp["exchange_inexact_frames"] == 0
    port = _free_port()
    outs = run(["--require-rccl"])
    assert [code for _, _, code in outs] == [3, 3], [o[1][-500:] for o in outs]
    assert "not over tr_allgather_frame" in outs[0][1]
