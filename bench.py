#!/usr/bin/env python3
"""bench.py — the headline metric of BASELINE.json on MI355X.

metric   shaded Mpixels/s (+ p50 frame ms) of the 4K transmissive pass: `fragment_transmission`
         (shader/src/lib.rs:37-162) over a fully covered synthetic TGB-v1 G-buffer, DragonAttenuation's
         light rig (sun + 1 punctual light), RGBA16F target, inputs resident in HBM.
step     one transmissive pass over one frame (tr_shade_transmission on the current stream).
N > 1    one process per GPU (torch.distributed / RCCL for rendezvous + barriers only): screen row bands,
         weak scaling — every rank shades 3840x2160 = 8.29 Mpx of a frame that grows with N
         (N=2 3840x4320, N=4 7680x4320 (the 8K of BASELINE config 5), N=8 7680x8640); replicated read-only
         inputs (tables, LUT, opaque pyramid), no data-path collective inside the timed region.  The frame
         composite (RCCL all-gather of the bands, sharded.allgather_frame) is timed separately and reported
         as `composite_allgather_ms`; it is xGMI-bound and not part of `value`.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ALGORITHMIC_BYTES_PER_PIXEL = 60  # 44 B G-buffer read + 8 B opaque-colour read + 8 B RGBA16F write
READ_BYTES_PER_PIXEL = 52
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); ~6.3 TB/s achievable


def frame_size_for(n_gpus: int, base_w: int, base_h: int):
    """Weak scaling: per-rank pixel count fixed; the frame doubles in height, then width, then height."""
    w, h, k = base_w, base_h, n_gpus
    grow_h = True
    while k > 1:
        if grow_h:
            h *= 2
        else:
            w *= 2
        grow_h = not grow_h
        k //= 2
    return w, h


def make_mip0_torch(width: int, height: int, device) -> torch.Tensor:
    """Same procedural opaque frame as synthetic.make_opaque_mip0, evaluated on the device (bench input only)."""
    xs = (torch.arange(width, dtype=torch.float32, device=device) + 0.5)[None, :]
    ys = (torch.arange(height, dtype=torch.float32, device=device) + 0.5)[:, None]
    sq = max(width // 120, 2)
    checker = ((torch.div(xs, sq, rounding_mode="floor") + torch.div(ys, sq, rounding_mode="floor")) % 2)
    g = 0.25 + 0.75 * checker
    r = g * (0.2 + 1.8 * xs / width)
    gch = g * (0.2 + 1.8 * ys / height)
    bch = g * (1.0 + 0.8 * torch.sin(xs / width * 12.0) * torch.cos(ys / height * 9.0))
    spot = torch.exp(-(((xs / width - 0.3) ** 2 + (ys / height - 0.4) ** 2) * 900.0)) * 3.0
    spot = spot + torch.exp(-(((xs / width - 0.72) ** 2 + (ys / height - 0.63) ** 2) * 2500.0)) * 2.0
    img = torch.empty((height, width, 4), dtype=torch.float16, device=device)
    img[..., 0] = torch.clamp(r + spot, 0.0, 4.0)
    img[..., 1] = torch.clamp(gch + spot, 0.0, 4.0)
    img[..., 2] = torch.clamp(bch + spot, 0.0, 4.0)
    img[..., 3] = 1.0
    return img


def cpu_baseline(scene, lut, width, height, budget_s=12.0):
    """The oracle (scalar fp32 C restatement of the reference, row-band threads) timed on this host's cores on a
    bounded sample of the same workload: whole rows from the middle of the frame, sized from a short probe so
    the timed run costs about `budget_s` seconds."""
    from oracle import oracle
    from transmission_renderer_amd import synthetic
    cores = os.cpu_count() or 1
    binding = oracle.SceneBinding(scene, lut)
    tex = oracle.new_pyramid(width, height, synthetic.make_opaque_mip0(width, height))
    oracle.generate_mips(width, height, tex)
    hdr16 = np.zeros((height, width, 4), dtype=np.float16)
    hdr32 = np.zeros((height, width, 4), dtype=np.float32)

    def run(rows):
        y0 = max(0, height // 2 - rows // 2)
        band = synthetic.make_gbuffer(width, height, rows=(y0, y0 + rows))
        t0 = time.perf_counter()
        oracle.shade_transmission(binding, band, tex, hdr_f16=hdr16, hdr_f32=hdr32, nthreads=cores)
        return time.perf_counter() - t0

    run(min(height, max(cores, 16)))                 # warm-up (page in the planes, spin up the threads)
    dt1 = run(height)                                # one whole frame
    reps = max(1, min(200, int(budget_s / max(dt1, 1e-3))))   # many-core hosts finish a frame in < 1 s
    rows = height
    dt = dt1 + sum(run(rows) for _ in range(reps - 1))
    rows_total = rows * reps
    return {"value": rows_total * width / dt / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"oracle/libtr_oracle.so o_shade_transmission, {reps} x {rows} rows x {width} px of the same "
                      f"frame ({rows_total * width / 1e6:.1f} Mpx) in {dt:.1f} s, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lights", type=int, default=1, help="punctual lights besides the sun (DragonAttenuation: 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from transmission_renderer_amd import synthetic, wire
    from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut

    fw, fh = frame_size_for(world, args.width, args.height)
    band_rows = fh // world
    y0, y1 = rank * band_rows, (rank + 1) * band_rows

    r = TransmissionRenderer(local_rank)
    dev = r.device
    scene = synthetic.make_scene(fw, fh, num_point_lights=args.lights, with_gbuffer=False)
    lut = load_ggx_lut()
    r.upload_materials(scene["materials"])
    r.upload_lights(scene["lights"])
    r.upload_ggx_lut(lut)
    r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev),
                         torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
    band = synthetic.make_gbuffer(fw, fh, rows=(y0, y1))          # this rank's screen tile only
    g = GBufferPlanes.from_numpy(band, dev)
    pyr = OpaquePyramid(fw, fh, dev)                              # replicated read-only input
    pyr.level(0).copy_(make_mip0_torch(fw, fh, dev))
    r.generate_mips(pyr)
    hdr = torch.zeros((fh, fw, 4), dtype=torch.float16, device=dev)
    uniforms, push = scene["uniforms"], scene["push"]
    rect = (0, y0, fw, y1)
    torch.cuda.synchronize()

    def step():
        r.shade_transmission(g, uniforms, push, pyr, hdr, rect)

    # Untimed warm-up: W steps — and before them, as many as it takes to have kept the GPU busy for 50 ms: its clocks
    # ramp over the first ~10 ms of load (measured: the first ~70 back-to-back 4K launches run 10-15 % slow), and the
    # metric is steady-state throughput.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.05:
        for _ in range(16):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    # Timed region: exactly K launches back to back, bracketed by the barrier + synchronize pairs (wall clock -> value)
    # and by ONE pair of HIP events on the launch stream (-> the kernel's average launch duration for the roofline).
    # No per-launch events here: each record is a barrier packet between two launches and costs them ~5 us apiece.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    avg_launch_ms = ev0.elapsed_time(ev1) / args.steps
    # Second pass, outside the timed region: per-launch events for the frame-time percentiles of the metric.
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    for i in range(args.steps):
        starts[i].record()
        step()
        ends[i].record()
    torch.cuda.synchronize()
    kernel_ms = np.array([s.elapsed_time(e) for s, e in zip(starts, ends)], dtype=np.float64)

    composite_ms = None
    if distributed:
        from transmission_renderer_amd import sharded
        for _ in range(3):
            sharded.allgather_frame(hdr, world)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            sharded.allgather_frame(hdr, world)
        e1.record()
        torch.cuda.synchronize()
        composite_ms = e0.elapsed_time(e1) / 10.0

    pixels_rank = band_rows * fw
    pixels_total = pixels_rank * world
    ms_per_step = elapsed / args.steps * 1e3
    value = pixels_total * args.steps / elapsed / 1e6
    avg_kernel_s = avg_launch_ms * 1e-3
    achieved = pixels_rank * ALGORITHMIC_BYTES_PER_PIXEL / avg_kernel_s / 1e9

    if rank == 0:
        out = {
            "metric": "shaded Mpixels/sec, 4K transmissive pass (fragment_transmission over a synthetic TGB-v1 G-buffer)",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "p50_frame_ms": round(float(np.percentile(kernel_ms, 50)), 4),
            "p10_frame_ms": round(float(np.percentile(kernel_ms, 10)), 4),
            "p90_frame_ms": round(float(np.percentile(kernel_ms, 90)), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"transmissive pass, {args.width}x{args.height} px per GPU (frame {fw}x{fh}), "
                                   f"TGB-v1 synthetic G-buffer fully covered, 16 materials, sun + {args.lights} "
                                   f"punctual light(s) (DragonAttenuation rig), RGBA16F target, {pyr.levels}-level "
                                   f"opaque pyramid, ggx_lut.png",
                       "pixels_per_gpu": pixels_rank, "sharding": f"{world} row band(s) of {band_rows} rows"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "tr::shade_kernel<true, uint2>", "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": pixels_rank * ALGORITHMIC_BYTES_PER_PIXEL,
                         "read_only_frac": round(pixels_rank * READ_BYTES_PER_PIXEL / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4)},
        }
        traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(traffic_file):
            try:
                with open(traffic_file) as f:
                    tr_ = json.load(f)
                if tr_.get("width") == fw and tr_.get("height") == fh and tr_.get("lights") == args.lights and world == 1:
                    out["roofline"]["traffic"] = tr_["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = tr_.get("source")
            except Exception:
                pass
        if composite_ms is not None:
            out["composite_allgather_ms"] = round(composite_ms, 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, lut, fw, fh, args.cpu_budget_s)
            out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 3)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
