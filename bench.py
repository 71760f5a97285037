#!/usr/bin/env python3
"""bench.py — the headline metric of BASELINE.json on MI355X.

metric   shaded Mpixels/s (+ p50 frame ms) of the 4K transmissive pass: `fragment_transmission`
         (shader/src/lib.rs:37-162) over a fully covered synthetic TGB-v1 G-buffer, DragonAttenuation's
         light rig (sun + 1 punctual light), RGBA16F target, inputs resident in HBM.
step     one transmissive pass over one 3840x2160 frame (tr_shade_transmission) — at N > 1 followed by the
         composite (tr_allgather_frame: RCCL all-gather of the row bands; by default of the frame AS PRESENTED: every
         rank tonemaps its band to RGBA8 inside the step, `--composite-format rgba16f` gathers the HDR target instead).
one frame, two launches (N = 1)   `--split P` (default 2): a step shades the frame as P row bands (tr_band_rows: 4-row
         aligned), band i through its own tr_shade_transmission call (the sharding API's rect) on HIP stream i.  On ONE
         stream a launch waits for the last wave of the one before, and ~5 % of a launch is its fill and drain; with the
         frame in two half-frame launches on two streams the hardware starts one band's waves in the wave slots the other
         band's stragglers leave empty — disjoint halves of one frame, nothing shared, nothing cached (4K: 81.4 -> 76.9 us,
         tools/gpu_overlap_probe.py; four bands on four streams: 83.8).  `value` and `roofline` are wall clock / K of that
         region; a launch's own start-to-end duration (what rocprofv3 lists) is about one step, for half a frame.  The
         same K frames as whole-frame launches on one stream are timed right after and reported as `single_stream`.
frames in flight (N = 1)   `--streams S` (default 1): S > 1 issues step k on HIP stream k mod S, every frame in flight with
         its OWN copy of the G-buffer planes, the opaque pyramid and the colour target (and one launch per frame).  Two
         such frames gain nothing (83.8 vs 81.4 us: two frames' windows in every L2).  (Round 3's first bench ran two
         frames in flight over ONE shared G-buffer: the second frame found the first one's plane reads in the caches —
         75.8 us — which no renderer's consecutive frames do.  Those figures are withdrawn.)

python bench.py --gpus N   starts by itself: with WORLD_SIZE unset and N > 1 it spawns N child processes (one per
         GPU, before anything touches a GPU) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 set; under
         torch.distributed.run it uses the environment it is given.

N > 1    BASELINE config 4 / north_star: ONE 3840x2160 frame cut into N row bands (tr_band_rows: 4-row aligned), one
         process per GPU, replicated read-only inputs (tables, LUT, opaque pyramid), no collective while shading:
         STRONG scaling (`--scaling strong`, the default).  The timed step is band kernel + composite, pipelined the
         way a renderer would run it (`--composite overlap`: frame k's all-gather runs on a second stream under frame
         k+1's shading, two frame buffers); `value` = frame pixels x K / wall time of the K steps, max over ranks.
         Reported next to it: the kernel-only rate (no composite), the composite alone, rank 0's whole-frame
         single-GPU time measured in the same run and both speed-ups over it.
         `--scaling weak` keeps round 1's mode (every rank shades 8.29 Mpx of a frame that grows with N).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Bytes per shaded pixel.  SURVEY.md 8d counts 44 B G-buffer read + 8 B opaque-colour read + 8 B RGBA16F write = 60 B
# and names 52 B (its read-only figure) the conservative claim.  52 B is also exactly what THIS variant has to move: the
# untextured kernel never loads the 8 B/px uv plane (16 + 16 + 4 planes + 8 opaque colour + 8 write).  `roofline.frac`
# is priced on the 52; the 60 B figure is reported beside it as `frac_survey_60B`.
SURVEY_BYTES_PER_PIXEL = 60
ALGORITHMIC_BYTES_PER_PIXEL = 52
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); ~6.3 TB/s achievable


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lights", type=int, default=1, help="punctual lights besides the sun (DragonAttenuation: 1)")
    ap.add_argument("--roughness-override", type=float, default=None, help="BASELINE config 3 uses 0.25")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--composite", choices=("overlap", "serial", "none"), default="overlap",
                    help="N > 1: how the composite all-gather enters the timed step")
    ap.add_argument("--composite-format", choices=("rgba8", "rgba16f"), default="rgba8",
                    help="N > 1: what is composited — rgba8: the frame as it is presented (every rank tonemaps its band with "
                         "tr_tonemap inside the step, the 4 B/px bands are gathered); rgba16f: the HDR target itself (8 B/px)")
    ap.add_argument("--split", type=int, default=0,
                    help="N = 1: a step shades the frame as P row bands, band i on HIP stream i (1: one whole-frame launch; "
                         "0 = default: 2 for frames of 4 Mpixels and more, else 1 — at 1080p two small launches cost more than "
                         "their overlap gives: 31.3 vs 27.2 us)")
    ap.add_argument("--streams", type=int, default=1,
                    help="N = 1: frames in flight — step k goes to HIP stream k mod S, every frame in flight with its own inputs "
                         "and target (S > 1 implies --split 1)")
    ap.add_argument("--all-transmissive", action="store_true",
                    help="every synthetic material gets transmission_factor 1 (DragonAttenuation's is 1): no tile skips "
                         "the refraction taps; the default run reports this variant beside the headline number")
    ap.add_argument("--no-variants", action="store_true", help="N = 1: skip the extra reported variants")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="N = 1: do not measure roofline.traffic live (two short child runs of this script under "
                         "`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`); the committed profiles/pmc_traffic.json is used instead")
    ap.add_argument("--no-single-gpu-reference", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--ramp-s", type=float, default=None, help="seconds of back-to-back launches before the warm-up (clock ramp; "
                                                               "default 0.05, 0.15 with several frames in flight)")
    ap.add_argument("--launch-timeout-s", type=float, default=1500.0)
    ap.add_argument("--rehearse-distributed", action="store_true",
                    help="N = 1 only: run the N > 1 code path on one GPU (process group of one rank, the library's RCCL "
                         "communicator, comm stream, composite in the timed step) — every line of the multi-GPU path that "
                         "one GPU can execute")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="no GPU: the ranks rendezvous over gloo, cut the frame into bands and composite a host frame "
                         "(covers the launcher and the band arithmetic; tests/test_bench_launch.py)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(args) -> int:
    """N fresh processes, one per GPU.  This parent never touches a GPU (it does not even import torch) and never
    replaces itself: it waits for the children and returns the worst exit code."""
    n = args.gpus
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this pool)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    deadline = time.time() + args.launch_timeout_s
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed or time.time() > deadline:
            for i, p in enumerate(procs):      # the exact PIDs started above
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
            if not failed:
                sys.stderr.write(f"bench.py: ranks did not finish within {args.launch_timeout_s:.0f} s\n")
                return 124
            break
        time.sleep(0.05)
    return max(abs(c) for c in codes)


# ------------------------------------------------------------------------------------------------ helpers
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


def make_mip0_torch(width: int, height: int, device):
    """Same procedural opaque frame as synthetic.make_opaque_mip0, evaluated on the device (bench input only)."""
    import torch
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


def native_oracle():
    """The oracle as BASELINE.md 2 describes the CPU baseline: `-O3 -march=native`, no fast-math, no FMA contraction —
    compiled HERE (the host the baseline is timed on; a -march=native object from another machine may not even run)
    from oracle/tr_oracle.c into oracle/_native/.  The parity checker stays the portable -O2 build; the two give the same
    bits (no contraction, no reassociation).  Returns (ctypes library or None, description)."""
    import ctypes as C
    src = os.path.join(ROOT, "oracle", "tr_oracle.c")
    out_dir = os.path.join(ROOT, "oracle", "_native")
    out = os.path.join(out_dir, "libtr_oracle_native.so")
    flags = ["-O3", "-march=native", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared", "-pthread"]
    try:
        os.makedirs(out_dir, exist_ok=True)
        subprocess.run(["gcc"] + flags + [src, "-o", out, "-lm"], check=True, capture_output=True, timeout=120)
        return C.CDLL(out), "gcc " + " ".join(flags[:2])
    except Exception:
        return None, "gcc -O2 (oracle/Makefile; the -O3 -march=native build failed on this host)"


def cpu_baseline(scene, lut, width, height, budget_s=12.0):
    """The oracle (scalar fp32 C restatement of the reference, row-band threads) timed on this host's cores on a
    bounded sample of the same workload: whole frames, as many as fit the budget."""
    import numpy as np
    from oracle import oracle
    from transmission_renderer_amd import synthetic
    try:
        cores = len(os.sched_getaffinity(0))      # the cores this process may run on (cgroup / affinity), not the machine's
    except AttributeError:
        cores = os.cpu_count() or 1
    binding = oracle.SceneBinding(scene, lut)
    tex = oracle.new_pyramid(width, height, synthetic.make_opaque_mip0(width, height))
    oracle.generate_mips(width, height, tex)
    hdr16 = np.zeros((height, width, 4), dtype=np.float16)
    hdr32 = np.zeros((height, width, 4), dtype=np.float32)
    native, how = native_oracle()
    if native is not None:
        oracle._bind_passes(native)

    def run(rows):
        y0 = max(0, height // 2 - rows // 2)
        band = synthetic.make_gbuffer(width, height, rows=(y0, y0 + rows))
        t0 = time.perf_counter()
        oracle.shade_transmission(binding, band, tex, hdr_f16=hdr16, hdr_f32=hdr32, nthreads=cores, lib=native)
        return time.perf_counter() - t0

    run(min(height, max(cores, 16)))                 # warm-up (page in the planes, spin up the threads)
    dt1 = run(height)                                # one whole frame
    reps = max(1, min(200, int(budget_s / max(dt1, 1e-3))))   # many-core hosts finish a frame in < 1 s
    rows = height
    dt = dt1 + sum(run(rows) for _ in range(reps - 1))
    rows_total = rows * reps
    return {"value": rows_total * width / dt / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"oracle/tr_oracle.c o_shade_transmission ({how}), {reps} x {rows} rows x {width} px of the same "
                      f"frame ({rows_total * width / 1e6:.1f} Mpx) in {dt:.1f} s, {cores} threads"}


def measure_traffic(args):
    """HBM bytes per launch of the transmissive kernel from the PMC counters, measured NOW: two child runs of this script
    (a few untimed launches, one stream) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` — the
    counters in separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled (gfx950 tallies
    the 128-byte requests of 16 B/lane streaming reads at 64 B), KiB -> bytes.  Returns None when rocprofv3 is missing,
    this process already runs under it, or anything fails (the committed profiles/pmc_traffic.json is used then)."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="tr_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--streams", "1", "--split", "1", "--no-cpu-baseline",
                   "--no-variants", "--no-traffic", "--ramp-s", "0.0", "--width", str(args.width), "--height", str(args.height),
                   "--lights", str(args.lights)] + (["--roughness-override", str(args.roughness_override)] if args.roughness_override is not None else []) \
                  + (["--all-transmissive"] if args.all_transmissive else [])
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "shade_kernel<true" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
            if len(vals) < 4:
                return None
            vals.sort()
            out[counter] = vals[len(vals) // 2] * 1024.0      # median over the launches, KiB -> bytes
        return {"hbm_bytes_per_launch": int(round(2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"])),
                "fetch_size_bytes_raw": int(round(out["FETCH_SIZE"])), "write_size_bytes": int(round(out["WRITE_SIZE"])),
                "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate child "
                          "passes of this script (--steps 6 --split 1: whole-frame launches; median over the kernel's launches), KiB -> bytes; FETCH_SIZE "
                          "doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of 16 B/lane streaming reads at 64 B)"}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def frame_pipeline_time(width, height):
    """us per frame of tr_record_frame on the procedural `meshes` scene (own contexts; see run_rank): every frame behind
    the previous one, and with two frames in flight — two contexts (each its own work buffers and targets) on two HIP
    streams, frame k recorded into context k mod 2, like a renderer with a swapchain: a frame is a dependent chain of
    ~13 launches, five of them latency-bound, and a second frame fills their gaps."""
    import numpy as np
    import torch
    from transmission_renderer_amd import meshes, synthetic, wire
    from transmission_renderer_amd.renderer import TransmissionRenderer

    def make_context():
        r = TransmissionRenderer(0)
        scene = synthetic.make_scene(width, height, num_point_lights=2, with_gbuffer=False, textured=True)
        geometry = meshes.make_mesh_scene(extra_instances=True)
        scene["materials"][2].alpha_clipping_cutoff = 0.75
        scene["materials"][7].alpha_clipping_cutoff = 0.6
        r.upload_ggx_lut()
        r.upload_materials(scene["materials"])
        r.upload_textures(scene["textures"])
        r.upload_lights(scene["lights"])
        r.upload_geometry(geometry)
        _, view = wire.default_camera()
        aabbs = r.write_cluster_data(scene["uniforms"], wire.inverse_perspective(width, height), (width, height))
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(width, height), view)
        work = r.new_frame_buffers(width, height)
        q = wire.view_rotation_inverse(view)
        return r, (lambda: r.record_frame(scene["uniforms"], scene["push"], culling, view, q, aabbs, work)), len(geometry["index"]) // 3

    ctxs = []
    try:
        ctxs = [make_context(), make_context()]
        torch.cuda.synchronize()
        streams = [torch.cuda.current_stream(), torch.cuda.Stream()]

        def run(n, in_flight):
            for k in range(n):
                with torch.cuda.stream(streams[k % in_flight]):
                    ctxs[k % in_flight][1]()

        out = {}
        for in_flight in (1, 2):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2:
                run(8, in_flight)
                torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run(40, in_flight)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) / 40)
            out[in_flight] = float(np.median(ts))
        return {"us_per_frame": round(out[1] * 1e6, 1), "frames_per_s": round(1.0 / out[1], 1),
                "two_frames_in_flight": {"us_per_frame": round(out[2] * 1e6, 1), "frames_per_s": round(1.0 / out[2], 1),
                                         "note": "two contexts on two HIP streams, frame k recorded into context k mod 2"},
                "scene": f"procedural `meshes` scene, {ctxs[0][2]} triangles, textured + alpha-clipped + transmissive "
                         f"materials, sun + 2 punctual lights, {width}x{height}, RGBA16F + tonemapped RGBA8 out",
                "stages": "culling | light assignment -> demultiplex -> visibility-buffer rasteriser (2 layers) -> opaque -> "
                          "mip chain -> transmissive -> tonemap; one tr_record_frame call per frame, 320 frames back to back"}
    finally:
        for c in ctxs:
            c[0].close()


def selftest_cpu(args, world, rank):
    """Launcher + band arithmetic + composite order without a GPU (gloo, host tensors)."""
    import torch
    import torch.distributed as dist
    from transmission_renderer_amd import sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 96, args.height if args.height != 2160 else 54
    rows, y0, y1 = sharded.band_rows(h, world, rank)
    frame = torch.zeros((rows * world, w, 4), dtype=torch.float16)
    frame[y0:y1] = float(rank + 1)
    comp = sharded.Compositor(world, rank)
    comp.allgather_rows(frame)
    want = torch.zeros_like(frame)
    for r in range(world):
        _, a, b = sharded.band_rows(h, world, r)
        want[a:b] = float(r + 1)
    ok = bool(torch.equal(frame[:h], want[:h]))
    if world > 1:
        flag = torch.tensor([int(ok)])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": "ok" if ok else "FAILED", "n_ranks": world, "frame": [w, h], "rows_per_rank": rows,
                          "bands": [list(sharded.band_rows(h, world, r)[1:]) for r in range(world)],
                          "composite": comp.backend}), flush=True)
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args) -> int:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)
    import numpy as np
    import torch
    distributed = world > 1 or args.rehearse_distributed
    dist = None
    if distributed:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from transmission_renderer_amd import sharded, synthetic
    from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut

    K, W = args.steps, args.warmup
    strong = args.scaling == "strong"
    fw, fh = (args.width, args.height) if strong else frame_size_for(world, args.width, args.height)
    rows_per_rank, y0, y1 = sharded.band_rows(fh, world, rank)
    padded = rows_per_rank * world
    composite = args.composite if distributed else "none"

    r = TransmissionRenderer(local_rank)
    dev = r.device
    scene = synthetic.make_scene(fw, fh, num_point_lights=args.lights, with_gbuffer=False,
                                 roughness_override=args.roughness_override)
    if args.all_transmissive:
        for m in scene["materials"]:
            m.transmission_factor = 1.0
    lut = load_ggx_lut()
    r.upload_materials(scene["materials"])
    r.upload_lights(scene["lights"])
    r.upload_ggx_lut(lut)
    r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev),
                         torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
    rect_whole = (0, y0, fw, y1)
    g = GBufferPlanes.from_numpy(synthetic.make_gbuffer(fw, fh, rows=(y0, y1) if y1 > y0 else (0, 1)), dev)   # this rank's screen tile only
    pyr = OpaquePyramid(fw, fh, dev)                                                  # replicated read-only input
    pyr.level(0).copy_(make_mip0_torch(fw, fh, dev))
    r.generate_mips(pyr)
    n_streams = max(1, args.streams) if not distributed else 1
    split = (args.split if args.split > 0 else (2 if fw * fh >= 4_000_000 else 1)) if (not distributed and n_streams == 1) else 1
    parts = [(0, a, fw, b) for a, b in (sharded.band_rows(fh, split, i)[1:] for i in range(split)) if b > a] if split > 1 else [rect_whole]
    # every frame in flight has its OWN inputs as well as its own target — its own copy of the G-buffer planes and of the
    # opaque pyramid (consecutive frames of a renderer are different frames: one frame must not find the other's
    # plane reads in the caches)
    gs = [g] + [GBufferPlanes(g.pos_depth.clone(), g.nrm_scale.clone(), g.uv.clone(), g.material_id.clone(), g.origin_x, g.origin_y)
                for _ in range(n_streams - 1)]
    pyrs = [pyr]
    for _ in range(n_streams - 1):
        p2 = OpaquePyramid(fw, fh, dev)
        p2.texels.copy_(pyr.texels)
        pyrs.append(p2)
    frames = [torch.zeros((padded, fw, 4), dtype=torch.float16, device=dev)
              for _ in range(2 if composite == "overlap" else n_streams)]
    uniforms, push = scene["uniforms"], scene["push"]
    rect = (0, y0, fw, y1)
    # N > 1: what crosses the links.  rgba8 (default): the presented frame — the band is tonemapped (fragment_tonemap,
    # shader/src/lib.rs:683-697, what the reference's last pass does before the swapchain) inside the step and the 8-bit
    # bands are gathered: half the bytes per link of the RGBA16F target.
    present_ldr = distributed and composite != "none" and args.composite_format == "rgba8"
    ldr_frames = [torch.zeros((padded, fw, 4), dtype=torch.uint8, device=dev) for _ in frames] if present_ldr else None
    tonemap_params = r.baked_tonemap_params() if present_ldr else None
    comp = sharded.Compositor(world, rank, renderer=r, single_rank_comm=args.rehearse_distributed) if distributed else None
    torch.cuda.synchronize()

    compute = torch.cuda.current_stream()
    comm = torch.cuda.Stream() if composite == "overlap" else None
    flight = [compute] + [torch.cuda.Stream() for _ in range(max(n_streams, len(parts)) - 1)]   # N = 1: a stream per band / frame in flight
    shaded = [torch.cuda.Event() for _ in frames]
    gathered = [None for _ in frames]
    launches = {"count": 0}

    def shade(buf, slot=0):
        if y1 > y0:     # (a band can be empty when the height is far from a multiple of 4 N; it still joins the gathers)
            r.shade_transmission(gs[slot], uniforms, push, pyrs[slot], buf, rect)
        launches["count"] += 1

    def present(i):
        """The band as it is presented: tonemapped into this rank's rows of the 8-bit frame; returns what is gathered."""
        if not present_ldr:
            return frames[i]
        if y1 > y0:
            r.tonemap(frames[i][y0:y1], tonemap_params, out=ldr_frames[i][y0:y1])
        return ldr_frames[i]

    def step(k):
        """One step of the metric: this rank's band of the frame (+ the composite, N > 1)."""
        i = k % len(frames)
        buf = frames[i]
        if composite == "overlap":
            if gathered[i] is not None:
                compute.wait_event(gathered[i])        # the buffer's previous composite has left it
            shade(buf)
            out_buf = present(i)
            shaded[i].record(compute)
            with torch.cuda.stream(comm):
                comm.wait_event(shaded[i])
                comp.allgather_rows(out_buf)
                ev = torch.cuda.Event()
                ev.record(comm)
                gathered[i] = ev
        elif n_streams > 1:
            with torch.cuda.stream(flight[k % n_streams]):
                shade(buf, k % n_streams)
        elif len(parts) > 1:
            for i, part in enumerate(parts):       # the frame's row bands, each on its own stream
                with torch.cuda.stream(flight[i]):
                    r.shade_transmission(g, uniforms, push, pyr, buf, part)
                    launches["count"] += 1
        else:
            shade(buf)
            if composite == "serial":
                comp.allgather_rows(present(i))

    # Untimed warm-up: W steps — and before them, as many band launches as it takes to have kept the GPU busy for
    # 50 ms: its clocks ramp over the first ~10 ms of load (the first ~70 back-to-back 4K launches run 10-15 % slow),
    # and the metric is steady-state throughput.  The count is reported (`clock_ramp_launches`).
    # (with several frames in flight the ramp runs the same pattern as the timed region — the first few hundred
    #  launches after a second hardware queue comes into use run ~6 % slow, tools/gpu_overlap_probe.py — for 150 ms)
    t_ramp = time.perf_counter()
    ramp_k = 0
    multi = n_streams > 1 or len(parts) > 1
    while time.perf_counter() - t_ramp < (args.ramp_s if args.ramp_s is not None else (0.15 if multi else 0.05)):
        for _ in range(16):
            if multi:
                step(ramp_k)
                ramp_k += 1
            else:
                shade(frames[0])
        torch.cuda.synchronize()
    clock_ramp_launches = launches["count"]
    for k in range(W):
        step(k)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    # Timed region: exactly K steps back to back, bracketed by the barrier + synchronize pairs (wall clock -> value).
    # At N = 1 ONE pair of HIP events on the launch stream brackets the K launches (-> the kernel's average launch
    # duration for the roofline); no per-launch events: each record is a barrier packet that costs ~5 us.
    # With S > 1 streams every stream gets its own pair; the streams were synchronised just above, so the region runs
    # from the earliest start to the latest end: region = (the first stream's start event) -> (every stream's end event).
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in flight]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in flight]
    t0 = time.perf_counter()
    for s_, e in zip(flight, ev0):
        e.record(s_)
    for k in range(K):
        step(W + k)
    for s_, e in zip(flight, ev1):
        e.record(s_)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    region_ms = max(ev0[0].elapsed_time(e) for e in ev1) / K
    # a launch's own time on its stream (what a per-kernel trace shows: launches of different streams overlap)
    in_stream_ms = max(a.elapsed_time(b) / max(1, K if len(parts) > 1 else len(range(i, K, n_streams)))
                       for i, (a, b) in enumerate(zip(ev0, ev1)))

    # ---- outside the timed region -------------------------------------------------------------------------------
    def timed_launches(n, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    # the band kernel alone, back to back (at N = 1 the timed region already is exactly this)
    kernel_ms = region_ms if not distributed else timed_launches(K, lambda: shade(frames[0]))
    # ... and with two frames in flight on two streams, like the N = 1 metric: a band of 1/N frame is mostly its own fill
    # and drain (tools/gpu_band_timing.py), which the next frame's band overlaps
    kernel_in_flight_ms = None
    if distributed and len(frames) >= 2:
        ko = [compute, torch.cuda.Stream()]
        if len(gs) < 2:      # (the second frame in flight reads its own copy of the band's planes and of the pyramid)
            gs.append(GBufferPlanes(g.pos_depth.clone(), g.nrm_scale.clone(), g.uv.clone(), g.material_id.clone(), g.origin_x, g.origin_y))
            p2 = OpaquePyramid(fw, fh, dev)
            p2.texels.copy_(pyr.texels)
            pyrs.append(p2)

        def two_in_flight(n):
            for k in range(n):
                with torch.cuda.stream(ko[k % 2]):
                    shade(frames[k % 2], k % 2)
            compute.wait_stream(ko[1])
        timed_launches(1, lambda: two_in_flight(32))
        kernel_in_flight_ms = timed_launches(1, lambda: two_in_flight(K)) / K
    # N = 1: the same K launches on ONE stream, each behind the previous one (how rounds 1 and 2 ran the metric)
    single_stream_ms = timed_launches(K, lambda: shade(frames[0])) if (not distributed and multi) else None
    kernel_ms_max = kernel_ms
    composite_ms = None
    per_rank_kernel_ms = [kernel_ms]
    if distributed:
        mine = torch.tensor([kernel_ms, float((y1 - y0) * fw)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_kernel_ms = [float(e[0].item()) for e in every]
        per_rank_pixels = [float(e[1].item()) for e in every]
        kernel_ms_max = max(per_rank_kernel_ms)
        if kernel_in_flight_ms is not None:
            t = torch.tensor([kernel_in_flight_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            kernel_in_flight_ms = float(t.item())
        dist.barrier()
        composite_ms = timed_launches(max(10, K // 4), lambda: comp.allgather_rows(frames[0]))
        t = torch.tensor([composite_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        composite_ms = float(t.item())
        # the same composite of the frame as it is presented: bands tonemapped to RGBA8 first (half the bytes per link)
        ldr = torch.zeros((padded, fw, 4), dtype=torch.uint8, device=dev)
        dist.barrier()
        composite_ldr_ms = timed_launches(max(10, K // 4), lambda: comp.allgather_rows(ldr))
        t = torch.tensor([composite_ldr_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        composite_ldr_ms = float(t.item())
    # per-launch events for the frame-time percentiles of the metric
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    for i in range(K):
        starts[i].record()
        shade(frames[0])
        ends[i].record()
    torch.cuda.synchronize()
    per_launch_ms = np.array([s.elapsed_time(e) for s, e in zip(starts, ends)], dtype=np.float64)
    # a launch + synchronise per step (what a host loop that waits for every frame sees: the GPU never reaches its
    # steady-state clocks in this pattern)
    sync_ms = []
    for _ in range(50):
        t1 = time.perf_counter()
        shade(frames[0])
        torch.cuda.synchronize()
        sync_ms.append((time.perf_counter() - t1) * 1e3)
    lps = len(parts) if (not distributed and n_streams == 1) else 1      # kernel launches per step
    launch_log = [("clock_ramp", clock_ramp_launches), ("warmup", W * lps), ("timed", K * lps)]
    if distributed:
        launch_log.append(("kernel_only", K))
    if single_stream_ms is not None:
        launch_log.append(("single_stream", K))
    launch_log += [("percentiles", K), ("launch_sync", 50)]

    pixels_rank = (y1 - y0) * fw
    pixels_frame = fw * fh                     # strong: all ranks together shade one frame per step
    pixels_step = pixels_frame if strong else pixels_rank * world
    value = pixels_step * K / elapsed / 1e6
    ms_per_step = elapsed / K * 1e3

    single_gpu_ms = None
    single_gpu_in_flight_ms = None
    if distributed and strong and not args.no_single_gpu_reference:
        if rank == 0:                          # the same frame on ONE GPU, same run: the denominator of the speed-ups
            gw = GBufferPlanes.from_numpy(synthetic.make_gbuffer(fw, fh), dev)
            whole = torch.zeros((fh, fw, 4), dtype=torch.float16, device=dev)
            fn = lambda: r.shade_transmission(gw, uniforms, push, pyr, whole)   # noqa: E731
            t_r = time.perf_counter()      # (the GPU idled while the host made the planes: ramp its clocks again)
            while time.perf_counter() - t_r < 0.05:
                timed_launches(16, fn)
            single_gpu_ms = timed_launches(K, fn)
            if kernel_in_flight_ms is not None:
                whole2 = torch.zeros((fh, fw, 4), dtype=torch.float16, device=dev)
                gw2 = GBufferPlanes(gw.pos_depth.clone(), gw.nrm_scale.clone(), gw.uv.clone(), gw.material_id.clone())
                ko2 = [compute, torch.cuda.Stream()]

                def whole_in_flight(n):
                    for k in range(n):
                        with torch.cuda.stream(ko2[k % 2]):
                            r.shade_transmission(gw if k % 2 == 0 else gw2, uniforms, push, pyrs[k % 2], whole if k % 2 == 0 else whole2)
                    compute.wait_stream(ko2[1])
                timed_launches(1, lambda: whole_in_flight(64))
                single_gpu_in_flight_ms = timed_launches(1, lambda: whole_in_flight(K)) / K
                del whole2, gw2
            del gw, whole
        dist.barrier()

    variants = {}
    if not distributed and not args.no_variants and not args.all_transmissive:
        # DragonAttenuation's material has transmission_factor 1: no tile skips the refraction taps and the btdf lobes
        for m in scene["materials"]:
            m.transmission_factor = 1.0
        r.upload_materials(scene["materials"])
        def in_flight(n):
            for k in range(n):
                step(k)
            for s_ in flight[1:]:
                compute.wait_stream(s_)
        timed_launches(200, lambda: shade(frames[0]))
        ms = timed_launches(1, lambda: in_flight(K)) / K
        launch_log += [("all_transmissive_ramp", 200), ("all_transmissive", K * lps)]
        variants["all_transmissive"] = {
            "avg_kernel_ms": round(ms, 4), "Mpixels_per_s": round(pixels_rank / ms / 1e3, 1),
            "frac": round(pixels_rank * ALGORITHMIC_BYTES_PER_PIXEL / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_survey_60B": round(pixels_rank * SURVEY_BYTES_PER_PIXEL / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "streams": n_streams,
            "note": "every synthetic material with transmission_factor = 1 (in the headline scene 4 of 16 have 0 and skip "
                    "the refraction taps, LUT and btdf lobes)"}

    # The whole frame of the path this pass belongs to, for the record (reported, not the metric): culling -> light
    # assignment -> demultiplex -> rasteriser -> opaque -> mip chain -> transmissive -> tonemap through ONE native call per
    # frame (tr_record_frame) on a procedural textured scene at the same frame size, back to back.
    frame_pipeline = None
    if not distributed and not args.no_variants and not args.all_transmissive and args.roughness_override is None:
        try:
            frame_pipeline = frame_pipeline_time(fw, fh)
        except Exception as e:   # (never costs the metric its line)
            frame_pipeline = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        kernel_s = kernel_ms * 1e-3
        achieved = pixels_rank * ALGORITHMIC_BYTES_PER_PIXEL / kernel_s / 1e9
        out = {
            "metric": "shaded Mpixels/sec, 4K transmissive pass (fragment_transmission over a synthetic TGB-v1 G-buffer)",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(ms_per_step, 4), "p50_frame_ms": round(float(np.percentile(per_launch_ms, 50)), 4),
            "p10_frame_ms": round(float(np.percentile(per_launch_ms, 10)), 4),
            "p90_frame_ms": round(float(np.percentile(per_launch_ms, 90)), 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"transmissive pass, frame {fw}x{fh}"
                                   + (f" in {world} row bands of {rows_per_rank} rows" if distributed else "")
                                   + f", TGB-v1 synthetic G-buffer fully covered, 16 materials"
                                   + (" (all transmission_factor 1)" if args.all_transmissive else "")
                                   + (f", roughness override {args.roughness_override}" if args.roughness_override is not None else "")
                                   + f", sun + {args.lights} punctual light(s) (DragonAttenuation rig), RGBA16F target, "
                                     f"{pyr.levels}-level opaque pyramid, ggx_lut.png"
                                   + (f", {n_streams} frames in flight (each on its own HIP stream, with its own G-buffer planes, opaque pyramid and colour target)"
                                      if n_streams > 1 else "")
                                   + (f", every frame shaded as {len(parts)} row bands (one tr_shade_transmission call each) on {len(parts)} HIP streams"
                                      if lps > 1 else ""),
                       "pixels_per_step": pixels_step, "pixels_per_gpu": pixels_rank,
                       "sharding": f"{world} row band(s) of {rows_per_rank} rows (tr_band_rows)",
                       "composite": ("none (one GPU holds the frame)" if not distributed else
                                     f"{composite}: {comp.backend}"
                                     + ("" if composite == "none" else
                                        ", of the frame as presented: every rank tonemaps its band (tr_tonemap) inside the step and "
                                        "the RGBA8 bands (4 B/px) are gathered" if present_ldr else
                                        ", of the RGBA16F HDR target (8 B/px)"))},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "tr::shade_kernel<true, uint2, 0, false>", "avg_kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_pixel": ALGORITHMIC_BYTES_PER_PIXEL,
                         "algorithmic_bytes_per_launch": pixels_rank * ALGORITHMIC_BYTES_PER_PIXEL,
                         "frac_survey_60B": round(pixels_rank * SURVEY_BYTES_PER_PIXEL / kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                         "streams": n_streams, "launches_per_step": lps,
                         "kernel_ms_on_its_stream": round(in_stream_ms, 4),
                         "note": "52 B/px = what this untextured variant moves (16 + 16 + 4 B planes, 8 B opaque colour, 8 B "
                                 "write; = SURVEY 8d's read-only figure); SURVEY 8d's 60 B/px also counts the 8 B/px uv plane, "
                                 "which this kernel never loads (frac_survey_60B).  avg_kernel_ms = timed region / K = the time "
                                 f"of a step: {lps} launch(es) of the kernel, one per row band of the frame, each on its own stream; "
                                 "kernel_ms_on_its_stream = a launch's own duration as a per-kernel trace sees it (the bands' "
                                 "launches overlap: about one step for 1 / launches_per_step of the frame's bytes)"},
            "clock_ramp_launches": clock_ramp_launches,
            "launch_sync_p50_ms": round(float(np.percentile(sync_ms, 50)), 4),
            "launch_log": launch_log,
        }
        tr_ = None
        if world == 1 and not distributed and not args.no_traffic:
            tr_ = measure_traffic(args)            # live PMC passes (child processes; the GPU is idle here)
        traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if tr_ is None and os.path.exists(traffic_file) and world == 1 and not args.all_transmissive and args.roughness_override is None:
            try:
                with open(traffic_file) as f:
                    committed = json.load(f)
                if committed.get("width") == fw and committed.get("height") == fh and committed.get("lights") == args.lights:
                    tr_ = committed
            except Exception:
                pass
        if tr_ is not None:
            out["roofline"]["traffic"] = tr_["hbm_bytes_per_launch"]
            out["roofline"]["achieved_from_traffic"] = round(tr_["hbm_bytes_per_launch"] / kernel_s / 1e9, 1)
            out["roofline"]["frac_from_traffic"] = round(tr_["hbm_bytes_per_launch"] / kernel_s / 1e9 / HBM_PEAK_GBS, 4)
            out["roofline"]["traffic_source"] = tr_.get("source")
        if single_stream_ms is not None:
            ss = single_stream_ms * 1e-3
            out["single_stream"] = {"avg_kernel_ms": round(single_stream_ms, 4),
                                    "Mpixels_per_s": round(pixels_rank / single_stream_ms / 1e3, 1),
                                    "frac": round(pixels_rank * ALGORITHMIC_BYTES_PER_PIXEL / ss / 1e9 / HBM_PEAK_GBS, 4),
                                    "frac_survey_60B": round(pixels_rank * SURVEY_BYTES_PER_PIXEL / ss / 1e9 / HBM_PEAK_GBS, 4),
                                    "note": "the same K launches on one stream, each behind the previous one"}
        if distributed:
            out["kernel_only"] = {"ms_per_step": round(kernel_ms_max, 4),
                                  "Mpixels_per_s": round(pixels_step / kernel_ms_max / 1e3, 1),
                                  "per_rank_kernel_ms": [round(x, 4) for x in per_rank_kernel_ms],
                                  "per_rank_roofline_frac": [round(px_ * ALGORITHMIC_BYTES_PER_PIXEL / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                             if ms > 0 else None for ms, px_ in zip(per_rank_kernel_ms, per_rank_pixels)],
                                  "note": "the band kernels alone, max over ranks, no composite"}
            if kernel_in_flight_ms is not None:
                out["kernel_only"]["two_frames_in_flight"] = {
                    "ms_per_step": round(kernel_in_flight_ms, 4),
                    "Mpixels_per_s": round(pixels_step / kernel_in_flight_ms / 1e3, 1),
                    "note": "the same with consecutive frames' bands on two HIP streams (max over ranks)"}
            out["composite_allgather_ms"] = round(composite_ms, 4)
            out["composite_rgba8_allgather_ms"] = round(composite_ldr_ms, 4)
            if single_gpu_ms is not None:
                out["single_gpu_ms"] = round(single_gpu_ms, 4)
                out["speedup_vs_1gpu"] = {"kernel_only": round(single_gpu_ms / kernel_ms_max, 3),
                                          "with_composite": round(single_gpu_ms / ms_per_step, 3)}
                if kernel_in_flight_ms is not None:   # (both sides with two frames in flight)
                    out["speedup_vs_1gpu"]["kernel_only_two_frames_in_flight"] = round(single_gpu_in_flight_ms / kernel_in_flight_ms, 3)
        if variants:
            out["variants"] = variants
        if frame_pipeline:
            out["frame_pipeline"] = frame_pipeline
        if world == 1 and not args.no_cpu_baseline:
            if variants:   # the CPU baseline shades the headline scene
                scene = synthetic.make_scene(fw, fh, num_point_lights=args.lights, with_gbuffer=False,
                                             roughness_override=args.roughness_override)
            out["cpu_baseline"] = cpu_baseline(scene, lut, fw, fh, args.cpu_budget_s)
            out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 3)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        comp.close()
        dist.destroy_process_group()
    r.close()
    return 0


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_children(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        args.gpus = world
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
