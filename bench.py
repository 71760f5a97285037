#!/usr/bin/env python3
"""bench.py — the headline metric of BASELINE.json on MI355X.

metric   shaded Mpixels/s (+ p50 frame ms) of the 4K transmissive pass: `fragment_transmission`
         (shader/src/lib.rs:37-162) over a fully covered synthetic TGB-v1 G-buffer, DragonAttenuation's
         light rig (sun + 1 punctual light), RGBA16F target, inputs resident in HBM.
step     one transmissive pass over one 3840x2160 frame (tr_shade_transmission) — at N > 1 followed by the
         composite (tr_allgather_frame: RCCL all-gather of the row bands).
N = 1    (run_single) A step shades the frame as `--split P` row bands (default 2 from 4 Mpixels up), band i through its
         own tr_shade_transmission call on HIP stream i (the pattern INTEGRATION.md documents: one band's waves start in the
         slots the other band's stragglers leave).  COLD INPUTS: the timed steps rotate through `--sets` (default: as many
         as make > 1.2 GB, at least 3) distinct input sets — own G-buffer planes, own opaque pyramid, own target — so no
         step finds its inputs in the 256 MiB Infinity Cache; the same K steps over ONE set are reported beside it
         (`same_input`).  `value`, `ms_per_step` and `roofline.frac` all come from the wall clock of the K timed steps;
         the HIP-event bracket of the same region is reported as `roofline.frac_events`.
         The line also carries, each measured the same way in this run:
           configs.config2_1080p     BASELINE config 2's frame size: the transmissive pass, and opaque -> mips -> transmissive
                                     through `record`
           configs.config3_4k        4 punctual lights, roughness override 0.25 (arithmetic-bound: with a VALU roofline)
           configs.config5_8k_1gpu   config 5's 7680x4320 frame on one GPU
           configs.all_transmissive  every material transmission_factor 1 (what DragonAttenuation is)
         with us, Mpixels/s and the roofline fraction on 52 B/px and on SURVEY 8d's 60 B/px, and
           single_stream             one whole-frame tr_shade_transmission call per frame on one stream (rounds 1-2's step)
           frame_pipeline            tr_record_frame (culling -> rasteriser -> opaque -> mips -> transmissive -> tonemap), its
                                     launches one by one (`kernels`), and the same frame under OVERDRAW (`overdraw`: the
                                     objects in a closed room, 3 fragments per pixel, later draws nearer)
         roofline.traffic / roofline.valu: three short child runs of this script (`--pmc-probe`) under rocprofv3 --pmc
         (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_VALU ...: separate passes, kernel trace only).
python bench.py --gpus N   starts by itself: with WORLD_SIZE unset and N > 1 it spawns N child processes (one per
         GPU, before anything touches a GPU) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 set; under
         torch.distributed.run it uses the environment it is given.
N > 1    (run_rank) BASELINE config 4 / north_star: ONE 3840x2160 frame cut into N row bands (tr_band_rows: 4-row
         aligned), one process per GPU, replicated read-only inputs (tables, LUT, opaque pyramid), no collective while
         shading: STRONG scaling (`--scaling strong`, the default).  The timed step is band kernel + composite, pipelined
         the way a renderer would run it (`--composite overlap`: frame k's all-gather runs on a second stream under frame
         k+1's shading, two frame buffers); `value` = frame pixels x K / wall time of the K steps, max over ranks.
         Reported next to it: the kernel-only rate (no composite), the composite alone, rank 0's whole-frame
         single-GPU time measured in the same run and both speed-ups over it, and (`full_pipeline_8k`) the 8K frame of
         BASELINE config 5 through the sharded full pipeline — the one workload where sharding can pay.
         `--scaling weak` keeps round 1's mode (every rank shades 8.29 Mpx of a frame that grows with N).
         The line says which backend carried the composite (`composite_backend`, `rccl_ranks` = ncclCommCount of the
         library's communicator, `composite_fell_back`); `--require-rccl` exits with code 3 instead of measuring a
         fallback to torch.distributed.  `--one-device` rehearses this whole path on ONE GPU: every rank on cuda:0, a gloo
         process group, exchanges staged through host memory (RCCL refuses two ranks on one device) — the code path on
         device buffers, not the links (tests/test_gpu_two_ranks_one_device.py).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import signal
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
# The vector unit's measured issue peak (tools/ubench/valu_rate.hip on MI355X, profiles/r01/f_valu_issue_rates.txt): 1.12 G
# wave64 instructions/s per SIMD for dual-issued VGPR-operand fp32 mul/add/fma (2.14 cycles each at 2.4 GHz) x 64 lanes x
# 1024 SIMDs.  Instructions with an SGPR source, conversions / compares (4.0-4.1 cycles) and transcendentals (8.1) issue
# slower: a kernel's mix reaches less.
VALU_PEAK_LANEOPS = 1.12e9 * 64 * 1024


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
    ap.add_argument("--composite-format", choices=("rgb8", "rgba8", "rgba16f"), default="rgb8",
                    help="N > 1: what is composited — rgb8 (default): the frame as it is presented without its constant alpha "
                         "(every rank tonemaps its band with tr_tonemap_rgb8 inside the step, the 3 B/px bands are gathered); "
                         "rgba8: with the alpha byte (4 B/px, round 3); rgba16f: the HDR target itself (8 B/px)")
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
    ap.add_argument("--no-variants", action="store_true", help="N = 1: skip the extra reported configs / variants")
    ap.add_argument("--sets", type=int, default=0,
                    help="N = 1: distinct input sets (planes + pyramid + target) the timed steps rotate through; 0 = default: "
                         "as many as make more than 1.2 GB, at least 3")
    ap.add_argument("--frame-probe", action="store_true",
                    help="internal (the child runs under rocprofv3): 120 tr_record_frame calls of the 4K `meshes` scene, nothing else")
    ap.add_argument("--room", action="store_true", help="--frame-probe: the overdraw scene (meshes.make_mesh_scene(room=True))")
    ap.add_argument("--pmc-probe", action="store_true",
                    help="internal (the child runs under rocprofv3 --pmc): six whole-frame launches each of the headline "
                         "scene, the all-transmissive scene and config 3, nothing else")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="N = 1: do not measure roofline.traffic / roofline.valu live (three short child runs of this script "
                         "under `rocprofv3 --pmc ...`); the committed profiles/pmc_traffic.json is used for the traffic instead")
    ap.add_argument("--no-single-gpu-reference", action="store_true")
    ap.add_argument("--no-full-pipeline", action="store_true", help="N > 1: skip the 8K full-pipeline frame (full_pipeline_8k)")
    ap.add_argument("--full-pipeline-size", default="7680x4320",
                    help="N > 1: frame size of the sharded full pipeline (full_pipeline_8k; BASELINE config 5 is 7680x4320 — a "
                         "smaller one for the one-device rehearsal)")
    ap.add_argument("--exchange", choices=("halo", "allgather"), default="halo",
                    help="N > 1, full pipeline: how the opaque colour crosses the band borders between the passes — halo: rows "
                         "of levels 0 and 1 with the two neighbours + an all-gather of level 2 (1/16 of the bytes); allgather: "
                         "all of level 0 (round 3)")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--ramp-s", type=float, default=None, help="seconds of back-to-back launches before the warm-up (clock ramp; "
                                                               "default 0.05, 0.15 with several frames in flight)")
    ap.add_argument("--launch-timeout-s", type=float, default=1500.0)
    ap.add_argument("--rehearse-distributed", action="store_true",
                    help="N = 1 only: run the N > 1 code path on one GPU (process group of one rank, the library's RCCL "
                         "communicator, comm stream, composite in the timed step) — every line of the multi-GPU path that "
                         "one GPU can execute")
    ap.add_argument("--one-device", action="store_true",
                    help="N > 1 rehearsal on ONE GPU: every rank drives cuda:0, the process group is gloo and the exchanges of "
                         "device tensors are staged through host memory (RCCL refuses two ranks on one device).  Exercises "
                         "run_rank with WORLD_SIZE > 1 on device buffers — band origins, halo windows, the late verdict — not "
                         "the links: its numbers are not performance figures")
    ap.add_argument("--require-rccl", action="store_true",
                    help="N > 1: exit non-zero unless the composite ran over the library's own RCCL communicator "
                         "(tr_allgather_frame) — a silent fallback to torch.distributed would not be measuring it")
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


def make_gbuffer_torch(width: int, height: int, device, rows=None, num_materials: int = 16):
    """synthetic.make_gbuffer evaluated on the device (bench input only; the same analytic fields in float64, so the
    planes equal the host version's up to the last bit of libm vs the device's sin / cos): an 8K frame takes the host
    half a minute of numpy and the device a few milliseconds."""
    import math
    import torch
    from transmission_renderer_amd import wire
    from transmission_renderer_amd.renderer import GBufferPlanes
    f64 = torch.float64
    eye, view = wire.default_camera()
    proj = wire.perspective_matrix_reversed(width, height)
    y0, y1 = (0, height) if rows is None else rows
    xs = torch.arange(width, dtype=f64, device=device) + 0.5
    ys = torch.arange(y0, y1, dtype=f64, device=device) + 0.5
    ndc_x = (xs / width * 2.0 - 1.0)[None, :]
    ndc_y = (ys / height * 2.0 - 1.0)[:, None]
    fx, fy = float(proj[0, 0]), float(-proj[1, 1])
    dvx, dvy = ndc_x / fx, -ndc_y / fy
    zv = 2.6 + 0.6 * torch.sin(7.0 * ndc_x + 1.3) * torch.cos(5.0 * ndc_y) + 0.25 * torch.sin(23.0 * ndc_x * ndc_y)
    vx, vy, vz = dvx * zv, dvy * zv, -zv
    v = view.astype("float64")
    s_ = [float(v[0, 0]), float(v[1, 0]), float(v[2, 0])]
    u_ = [float(v[0, 1]), float(v[1, 1]), float(v[2, 1])]
    b_ = [float(v[0, 2]), float(v[1, 2]), float(v[2, 2])]
    e = [float(x) for x in eye]
    h = y1 - y0
    pos_depth = torch.empty((h, width, 4), dtype=torch.float32, device=device)
    nrm_scale = torch.empty((h, width, 4), dtype=torch.float32, device=device)
    for k in range(3):
        pos_depth[..., k] = e[k] + s_[k] * vx + u_[k] * vy + b_[k] * vz
    pa, pb = float(proj[2, 2]), float(proj[3, 2])
    pos_depth[..., 3] = (pa * vz + pb) / (-vz)
    inv = 1.0 / torch.sqrt(vx * vx + vy * vy + vz * vz)
    rx = 0.9 * torch.sin(31.0 * ndc_x + 2.0 * ndc_y) + 0.3 * torch.sin(97.0 * ndc_y)
    ry = 0.9 * torch.cos(27.0 * ndc_y - 3.0 * ndc_x) + 0.3 * torch.cos(89.0 * ndc_x)
    nvx, nvy, nvz = -vx * inv + rx, -vy * inv + ry, -vz * inv + 0.0 * rx
    nlen = 0.75 + 0.25 * torch.sin(11.0 * ndc_x + 5.0 * ndc_y)
    for k in range(3):
        nrm_scale[..., k] = (s_[k] * nvx + u_[k] * nvy + b_[k] * nvz) * nlen
    uv = torch.empty((h, width, 2), dtype=torch.float32, device=device)
    uv[..., 0] = (xs[None, :] / width * 4.0).expand(h, width)
    uv[..., 1] = (ys[:, None] / height * 4.0).expand(h, width)
    cw, ch = width / 16.0, height / 9.0
    xw = xs[None, :] + 0.35 * cw * torch.sin(ys[:, None] * (2.0 * math.pi / (3.1 * ch)))
    yw = ys[:, None] + 0.35 * ch * torch.sin(xs[None, :] * (2.0 * math.pi / (2.7 * cw)))
    cx = torch.floor(xw / cw).to(torch.int64)
    cy = torch.floor(yw / ch).to(torch.int64)
    hsh = (cx * 73856093) ^ (cy * 19349663) ^ ((cx + cy) * 83492791)
    hsh = (hsh ^ (hsh >> 13)) & 0x7FFFFFFF
    material_id = (hsh % num_materials).to(torch.int32)
    scale = torch.tensor([1.0, 0.5, 2.0, 1.0], dtype=torch.float32, device=device)
    nrm_scale[..., 3] = scale[(hsh >> 8) % 4]
    return GBufferPlanes(pos_depth, nrm_scale, uv, material_id.contiguous(), 0, y0)


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

    bands = {}

    def run(rows):
        if rows not in bands:                            # (the planes are inputs: made once, outside the timed region)
            y0 = max(0, height // 2 - rows // 2)
            bands[rows] = synthetic.make_gbuffer(width, height, rows=(y0, y0 + rows))
        t0 = time.perf_counter()
        oracle.shade_transmission(binding, bands[rows], tex, hdr_f16=hdr16, hdr_f32=hdr32, nthreads=cores, lib=native)
        return time.perf_counter() - t0

    run(min(height, max(cores, 16)))                 # warm-up (spin up the threads)
    run(height)                                      # ... and page in the whole frame's planes
    dt1 = run(height)                                # one whole frame
    reps = max(1, min(200, int(budget_s / max(dt1, 1e-3))))   # many-core hosts finish a frame in < 1 s
    rows = height
    dt = dt1 + sum(run(rows) for _ in range(reps - 1))
    rows_total = rows * reps
    return {"value": rows_total * width / dt / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"oracle/tr_oracle.c o_shade_transmission ({how}), {reps} x {rows} rows x {width} px of the same "
                      f"frame ({rows_total * width / 1e6:.1f} Mpx) in {dt:.1f} s, {cores} threads"}


PROBE_PHASES = ("headline", "all_transmissive", "config3")
PROBE_LAUNCHES = 6


def run_profiled(cmd, env, timeout):
    """A rocprofv3 child in its own session; on a timeout the WHOLE group is killed (killing only rocprofv3 left the profiled
    Python process running on the GPU beside whatever was timed next).  (returncode, stderr); returncode None: timed out."""
    p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        _, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)     # (the group this call started, by its id)
        except ProcessLookupError:
            pass
        p.communicate()
        return None, "timed out"
    return p.returncode, err


def measure_pattern_ceiling():
    """roofline.pattern_ceiling: transmission_renderer_amd/pattern_ceiling --brief (tools/ubench/pattern_ceiling.hip, built by
    __graft_entry__.build()) — a stand-alone HIP program that issues the pass's memory pattern on the pass's own tile numbering
    with NO shading arithmetic (two plane rows, ids, four 16-byte taps per pixel, a LUT line, the store; cold inputs): what
    this access pattern attains on this box in the same launch shapes.  None when the binary is missing or fails."""
    exe = os.path.join(ROOT, "transmission_renderer_amd", "pattern_ceiling")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "--brief"], capture_output=True, text=True, timeout=120)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if r.returncode == 0 and line else None
    except Exception:
        return None


def measure_pmc(args):
    """Counters of the transmissive kernel, measured NOW: three child runs of this script (`--pmc-probe`: PROBE_LAUNCHES
    whole-frame launches each of the headline scene, the all-transmissive scene and config 3, one stream) under
    `rocprofv3 --kernel-trace --pmc ...` — FETCH_SIZE, WRITE_SIZE and the SQ counters in separate passes, kernel trace
    only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of 16 B/lane
    streaming reads at 64 B), KiB -> bytes.  Returns {phase: {counter: median over the phase's launches}} or None when
    rocprofv3 is missing, this process already runs under it, or anything fails."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None
    out = {ph: {} for ph in PROBE_PHASES}
    tmp = tempfile.mkdtemp(prefix="tr_pmc_", dir="/tmp")
    try:
        for counters in (["FETCH_SIZE"], ["WRITE_SIZE"],
                         ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_WAIT_ANY"]):
            d = os.path.join(tmp, counters[0])
            cmd = [rocprof, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-probe", "--width", str(args.width), "--height", str(args.height),
                   "--lights", str(args.lights)]
            env = dict(os.environ, TMPDIR="/tmp")
            code, _ = run_profiled(cmd, env, 120)
            if code != 0:
                return None
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                rows += [row for row in csv.DictReader(open(f)) if "shade_kernel<true" in row["Kernel_Name"]]
            for c in counters:
                vals = [(int(row["Dispatch_Id"]), float(row["Counter_Value"])) for row in rows if row["Counter_Name"] == c]
                vals.sort()
                if len(vals) != PROBE_LAUNCHES * len(PROBE_PHASES):
                    return None
                for i, ph in enumerate(PROBE_PHASES):
                    seg = sorted(v for _, v in vals[i * PROBE_LAUNCHES:(i + 1) * PROBE_LAUNCHES])
                    out[ph][c] = seg[len(seg) // 2]
        for ph in PROBE_PHASES:
            c = out[ph]
            c["hbm_bytes_per_launch"] = int(round(2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0))
        out["source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU ... in "
                         "separate child passes of this script (--pmc-probe: whole-frame launches, median over each scene's "
                         f"{PROBE_LAUNCHES} launches), KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the "
                         "128-B requests of 16 B/lane streaming reads at 64 B)")
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_probe(args) -> int:
    """The child of measure_pmc: nothing but the launches (the parent finds them by dispatch order)."""
    import torch
    for phase in PROBE_PHASES:
        wl = PassWorkload(0, args.width, args.height, lights=4 if phase == "config3" else args.lights,
                          roughness=0.25 if phase == "config3" else None, all_transmissive=phase == "all_transmissive",
                          sets=1, split=1)
        for _ in range(PROBE_LAUNCHES):
            wl.step(0)
        torch.cuda.synchronize()
        wl.close()
    return 0


FRAME_PROBE_FRAMES = 120


def frame_probe(args) -> int:
    """The child of measure_frame_kernels: the frame recorder's loop of frame_pipeline_time, one context, nothing else."""
    import torch
    from transmission_renderer_amd import meshes, synthetic, wire
    from transmission_renderer_amd.renderer import TransmissionRenderer
    w, h = args.width, args.height
    r = TransmissionRenderer(0)
    scene = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False, textured=True)
    geometry = meshes.make_mesh_scene(extra_instances=True, room=args.room)
    scene["materials"][2].alpha_clipping_cutoff = 0.75
    scene["materials"][7].alpha_clipping_cutoff = 0.6
    r.upload_ggx_lut()
    r.upload_materials(scene["materials"])
    r.upload_textures(scene["textures"])
    r.upload_lights(scene["lights"])
    r.upload_geometry(geometry)
    _, view = wire.default_camera()
    aabbs = r.write_cluster_data(scene["uniforms"], wire.inverse_perspective(w, h), (w, h))
    culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    work = r.new_frame_buffers(w, h)
    q = wire.view_rotation_inverse(view)
    for _ in range(FRAME_PROBE_FRAMES):
        r.record_frame(scene["uniforms"], scene["push"], culling, view, q, aabbs, work)
    torch.cuda.synchronize()
    r.close()
    return 0


FRAME_KERNELS = (   # (name in the bench line, substring of the kernel's name in the trace)
    ("front (culling | light assignment | clear, demultiplex + draw scans behind it)", "frame_front_kernel"),
    ("set-up (vertex stage, edge functions, planes, item prefix)", "raster_setup_kernel"),
    ("rasteriser (both layers)", "raster_kernel"),
    ("opaque launch (visibility words; writes target, levels 0 and 1, presents)", "shade_kernel<false"),
    ("mip chain: even levels", "mip_even_kernel"),
    ("mip chain: first odd level", "downsample_kernel"),
    ("mip chain: tail", "mip_tail_kernel"),
    ("transmissive launch (visibility words, listed tiles; presents)", "shade_kernel<true"),
)


def measure_frame_kernels(width, height, room=False, counters=True):
    """frame_pipeline.kernels: the frame recorder's launches one by one, measured NOW by child runs of this script
    (`--frame-probe`) under rocprofv3 — one kernel-trace pass (µs per launch, launches per frame) and three --pmc passes
    (FETCH_SIZE; WRITE_SIZE; SQ counters), kernel trace only, as MI355X_MICROARCH.md prescribes.  Per kernel: us (mean
    over the frames behind the first 20), launches_per_frame, hbm_bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, median
    per launch), frac = hbm_bytes / us / 8 TB/s, issue_port_busy and waves_per_simd (see roofline.valu).  None when rocprofv3
    is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None
    tmp = tempfile.mkdtemp(prefix="tr_frame_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--frame-probe", "--width", str(width), "--height", str(height)] + (["--room"] if room else [])
    env = dict(os.environ, TMPDIR="/tmp")

    def rows_of(tag, extra):
        d = os.path.join(tmp, tag)
        code, err = run_profiled([rocprof, "--kernel-trace"] + extra + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + child, env, 240)
        if code != 0:
            raise RuntimeError(f"rocprofv3 {tag}: {err[-300:]}")
        return d

    try:
        d = rows_of("trace", [])
        per = {name: [] for name, _ in FRAME_KERNELS}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                for name, sub in FRAME_KERNELS:
                    if sub in row["Kernel_Name"]:
                        per[name].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
                        break
        out = {}
        for name, _ in FRAME_KERNELS:
            v = sorted(per[name])
            if not v:
                continue
            lpf = max(1, round(len(v) / FRAME_PROBE_FRAMES))
            steady = [dur for _, dur in v[20 * lpf:]] or [dur for _, dur in v]
            out[name] = {"us": round(sum(steady) / len(steady) / 1e3 * lpf, 2), "launches_per_frame": lpf}
        if not counters:
            return out
        counters = {}
        for tag, cs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                        ("sq", ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_INSTS_VALU"])):
            d = rows_of(tag, ["--pmc"] + cs)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    for name, sub in FRAME_KERNELS:
                        if sub in row["Kernel_Name"]:
                            counters.setdefault((name, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
                            break
        med = lambda name, c: (sorted(counters[(name, c)])[len(counters[(name, c)]) // 2] if (name, c) in counters else None)
        for name in out:
            k = out[name]
            fs, ws = med(name, "FETCH_SIZE"), med(name, "WRITE_SIZE")
            if fs is not None and ws is not None:
                k["hbm_bytes"] = int(round((2.0 * fs + ws) * 1024.0)) * k["launches_per_frame"]
                k["frac"] = round(k["hbm_bytes"] / (k["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            busy, act, act2, wc = (med(name, c) for c in ("SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_WAVE_CYCLES"))
            if busy:
                per_se = busy / 32.0
                k["issue_port_busy"] = round((act - act2) * 4.0 / 1024.0 / per_se, 3)
                k["waves_per_simd"] = round(wc * 4.0 / 1024.0 / per_se, 2)
                k["vector_instructions"] = int(med(name, "SQ_INSTS_VALU"))
        out["note"] = ("rocprofv3 child passes of `bench.py --frame-probe` (120 frames of the `meshes` scene, one context): us = mean "
                       "kernel time per frame behind the first 20 frames (kernel trace); hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE per "
                       "frame (separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md — calibrated on wide streaming "
                       "reads, an upper bound for the gather-heavy kernels); frac = hbm_bytes / us / 8 TB/s; issue_port_busy = "
                       "(SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) x 4 / 1024 SIMDs / (SQ_BUSY_CYCLES / 32)")
        return out
    except Exception as e:   # noqa: BLE001  (reported, never fatal for the bench line)
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class PassWorkload:
    """One configuration of the transmissive pass on one GPU: a context with its tables, `sets` distinct input sets
    (G-buffer planes, opaque pyramid, RGBA16F target) and the step that shades one frame — as `split` row bands, band i
    through its own tr_shade_transmission call on HIP stream i."""

    def __init__(self, device_index, width, height, lights=1, roughness=None, all_transmissive=False, sets=0, split=0):
        import numpy as np
        import torch
        from transmission_renderer_amd import sharded, synthetic
        from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut
        self.torch = torch
        self.w, self.h = width, height
        self.pixels = width * height
        self.r = r = TransmissionRenderer(device_index)
        dev = r.device
        self.scene = scene = synthetic.make_scene(width, height, num_point_lights=lights, with_gbuffer=False,
                                                  roughness_override=roughness)
        if all_transmissive:
            for m in scene["materials"]:
                m.transmission_factor = 1.0
        self.lut = load_ggx_lut()
        r.upload_materials(scene["materials"])
        r.upload_lights(scene["lights"])
        r.upload_ggx_lut(self.lut)
        r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev),
                             torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
        g = make_gbuffer_torch(width, height, dev)
        pyr = OpaquePyramid(width, height, dev)
        pyr.level(0).copy_(make_mip0_torch(width, height, dev))
        r.generate_mips(pyr)
        self.set_bytes = self.pixels * 44 + pyr.texels.numel() * 2 + self.pixels * 8
        n_sets = sets if sets > 0 else max(3, -(-1_200_000_000 // self.set_bytes) + 1)
        self.sets = [(g, pyr, torch.zeros((height, width, 4), dtype=torch.float16, device=dev))]
        for _ in range(n_sets - 1):
            p2 = OpaquePyramid(width, height, dev)
            p2.texels.copy_(pyr.texels)
            self.sets.append((GBufferPlanes(g.pos_depth.clone(), g.nrm_scale.clone(), g.uv.clone(), g.material_id.clone()), p2,
                              torch.zeros((height, width, 4), dtype=torch.float16, device=dev)))
        self.split = split if split > 0 else (2 if self.pixels >= 4_000_000 else 1)
        self.parts = [(0, a, width, b) for a, b in (sharded.band_rows(height, self.split, i)[1:] for i in range(self.split)) if b > a] \
            if self.split > 1 else [None]
        self.flight = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(len(self.parts) - 1)]
        self.levels = pyr.levels
        self.launches = 0
        torch.cuda.synchronize()

    def step(self, k, rotate=True):
        g, pyr, frame = self.sets[k % len(self.sets) if rotate else 0]
        u, pc = self.scene["uniforms"], self.scene["push"]
        if len(self.parts) == 1:
            self.r.shade_transmission(g, u, pc, pyr, frame, self.parts[0])
        else:
            for s_, part in zip(self.flight, self.parts):
                with self.torch.cuda.stream(s_):
                    self.r.shade_transmission(g, u, pc, pyr, frame, part)
        self.launches += len(self.parts)

    def whole_frame(self, k, rotate=True):
        g, pyr, frame = self.sets[k % len(self.sets) if rotate else 0]
        self.r.shade_transmission(g, self.scene["uniforms"], self.scene["push"], pyr, frame)
        self.launches += 1

    def ramp(self, seconds):
        """Back-to-back steps until the GPU has been busy for `seconds`: its clocks ramp over the first ~10 ms of load."""
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(16):
                self.step(k)
                k += 1
            self.torch.cuda.synchronize()

    def timed(self, K, fn=None, rotate=True, first=0):
        """K steps back to back: (wall-clock ms per step between two device synchronisations, HIP-event ms per step: from
        the first stream's start event to the last stream's end event)."""
        torch = self.torch
        fn = fn or self.step
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in self.flight]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in self.flight]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_, e in zip(self.flight, ev0):
            e.record(s_)
        for k in range(K):
            fn(first + k, rotate)
        for s_, e in zip(self.flight, ev1):
            e.record(s_)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / K * 1e3
        return wall, max(ev0[0].elapsed_time(e) for e in ev1) / K

    def figures(self, ms):
        return {"us": round(ms * 1e3, 2), "Mpixels_per_s": round(self.pixels / ms / 1e3, 1),
                "frac_52B": round(self.pixels * ALGORITHMIC_BYTES_PER_PIXEL / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_60B": round(self.pixels * SURVEY_BYTES_PER_PIXEL / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    def close(self):
        self.sets = []
        self.r.close()


def valu_roofline(counters, pixels, ms):
    """roofline.valu from the SQ counters of one launch and a step time: vector lane-operations against the vector unit's
    measured issue peak."""
    if not counters or "SQ_INSTS_VALU" not in counters:
        return None
    lane_ops = counters["SQ_INSTS_VALU"] * 64.0
    achieved = lane_ops / (ms * 1e-3)
    out = {"bound": "valu", "lane_ops_per_pixel": round(lane_ops / pixels, 1), "achieved": round(achieved / 1e12, 2),
           "peak": round(VALU_PEAK_LANEOPS / 1e12, 2), "unit": "T lane-ops/s", "frac": round(achieved / VALU_PEAK_LANEOPS, 4),
           "vector_instructions_per_launch": int(counters["SQ_INSTS_VALU"]),
           "scalar_instructions_per_launch": int(counters.get("SQ_INSTS_SALU", 0))}
    if all(k in counters for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES")):
        # SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles summed over the waves; SQ_BUSY_CYCLES cycles summed over the 32
        # shader engines.  The vector unit's first issue port is occupied by every instruction that is not the second half
        # of a dual-issued pair.
        busy = counters["SQ_BUSY_CYCLES"] / 32.0
        out["issue_port_busy"] = round((counters["SQ_ACTIVE_INST_VALU"] - counters["SQ_ACTIVE_INST_VALU2"]) * 4.0 / 1024.0 / busy, 3)
        out["dual_issued_share"] = round(counters["SQ_ACTIVE_INST_VALU2"] / counters["SQ_INSTS_VALU"], 3)
        out["waves_per_simd"] = round(counters["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / busy, 2)
        if "SQ_WAIT_ANY" in counters:
            out["wave_time_in_s_waitcnt"] = round(counters["SQ_WAIT_ANY"] / counters["SQ_WAVE_CYCLES"], 3)
    return dict(out, **{
            "note": "SQ_INSTS_VALU x 64 lanes / step time; peak = dual-issued VGPR-operand fp32 ops, 1.12 G wave64 "
                    "instructions/s/SIMD x 64 x 1024 SIMDs (tools/ubench/valu_rate.hip, profiles/r01/f_valu_issue_rates.txt); "
                    "an SGPR source, a conversion / compare (4.0-4.1 cycles) or a transcendental (8.1) issues slower, so the "
                    "kernel's instruction mix cannot reach 1.0.  issue_port_busy: the fraction of the launch's cycles in which a "
                    "SIMD's vector issue port was occupied ((SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) x 4 / 1024 SIMDs / (SQ_BUSY_CYCLES "
                    "/ 32)) — how close the launch (as profiled: one whole-frame launch per frame) is to the vector unit's roof"})


def measure_config(name, device_index, width, height, K, W, lights=1, roughness=None, all_transmissive=False, with_record=False):
    """One entry of the `configs` block: the transmissive pass of the configuration, steps as in the headline (row bands
    on streams, rotating cold input sets), the same steps over one input set, and a whole-frame launch per frame."""
    import torch
    wl = PassWorkload(device_index, width, height, lights=lights, roughness=roughness, all_transmissive=all_transmissive)
    try:
        wl.ramp(0.1)
        if W > 0:
            wl.timed(W)
        ms, _ = wl.timed(K)
        same, _ = wl.timed(K, rotate=False)
        single, _ = wl.timed(K, fn=wl.whole_frame)
        out = dict(wl.figures(ms), workload=f"transmissive pass {width}x{height}, sun + {lights} punctual light(s)"
                                           + (f", roughness override {roughness}" if roughness is not None else "")
                                           + (", every material transmission_factor 1" if all_transmissive else ""),
                   launches_per_step=len(wl.parts), input_sets=len(wl.sets), input_set_MB=round(wl.set_bytes / 1e6, 1),
                   same_input_us=round(same * 1e3, 2), single_stream_us=round(single * 1e3, 2),
                   single_stream_frac_52B=wl.figures(single)["frac_52B"])
        if with_record:
            # BASELINE config 2 "opaque + transmissive passes end-to-end": opaque -> mips -> transmissive through `record`
            # (the synthetic layer serves as both the opaque and the transmissive layer, as in smoke())
            def frame(k, rotate=True):
                g, pyr, hdr = wl.sets[k % len(wl.sets) if rotate else 0]
                wl.r.record(g, g, wl.scene["uniforms"], wl.scene["push"], hdr, pyr)
            wl.timed(max(W, 1), fn=frame)
            fms, _ = wl.timed(K, fn=frame)
            out["opaque_mips_transmissive_frame"] = {"us": round(fms * 1e3, 2), "Mpixels_per_s": round(wl.pixels / fms / 1e3, 1),
                                                     "note": "tr_shade_opaque + tr_generate_mips + tr_shade_transmission per frame, "
                                                             "one stream, rotating input sets"}
        return out
    finally:
        wl.close()
        torch.cuda.empty_cache()


def fragments_per_pixel(room, width=960, height=540):
    """Depth complexity of the mesh scene as the rasteriser meets it: every instance drawn ALONE through tr_draw_scene, its
    covered pixels counted (front-facing, un-clipped fragments), summed and divided by the frame's pixels."""
    import numpy as np
    import torch
    from transmission_renderer_amd import meshes, synthetic, wire
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    try:
        scene = synthetic.make_scene(width, height, num_point_lights=1, with_gbuffer=False, textured=True)
        scene["materials"][2].alpha_clipping_cutoff = 0.75
        scene["materials"][7].alpha_clipping_cutoff = 0.6
        r.upload_ggx_lut()
        r.upload_materials(scene["materials"])
        r.upload_textures(scene["textures"])
        _, view = wire.default_camera()
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(width, height), view)
        o, t = r.new_layer(width, height), r.new_layer(width, height)
        fragments = 0
        for mesh, draw_buffer, instances in meshes.mesh_scene_parts(True, room):
            for inst in instances:
                mb = meshes.ModelBuffers()
                mb.add_primitive(mesh, draw_buffer, [inst])
                r.upload_geometry(mb.finish())
                r.draw_scene(culling, scene["push"], o, t)
                fragments += int((o.material_id != -1).sum().item()) + int((t.material_id != -1).sum().item())
        return fragments / float(width * height)
    finally:
        r.close()


def frame_pipeline_time(width, height, room=False):
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
        geometry = meshes.make_mesh_scene(extra_instances=True, room=room)
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


def full_pipeline_8k(r, comp, world, rank, dev, dist, scene4k, K, args):
    """ms per frame of sharded.record_sharded on the 7680x4320 synthetic frame, max over ranks; rank 0 also times the same
    frame through `record` on one GPU."""
    import torch
    from transmission_renderer_amd import sharded, synthetic
    from transmission_renderer_amd.renderer import OpaquePyramid
    fw, fh = (int(v) for v in args.full_pipeline_size.lower().split("x"))
    scene = synthetic.make_scene(fw, fh, num_point_lights=args.lights, with_gbuffer=False)     # (same materials and lights: uploaded)
    uniforms, push = scene["uniforms"], scene["push"]
    rows, y0, y1 = sharded.band_rows(fh, world, rank)
    padded = rows * world
    g = make_gbuffer_torch(fw, fh, dev, rows=(y0, y1) if y1 > y0 else (0, 1))
    pyr = OpaquePyramid(fw, fh, dev, level0_rows=padded)
    hdr = torch.zeros((padded, fw, 4), dtype=torch.float16, device=dev)
    exchange = args.exchange if world > 1 else "allgather"

    def frame(confirm="late"):
        # (the halo's excess word is judged a frame late: no drain between a frame's passes and its composite; the frames in
        #  front of the timed ones settle the halo with the word read at once, the timed ones must all be confirmed exact)
        sharded.record_sharded(r, g, g, uniforms, push, hdr, pyr, comp, exchange=exchange, confirm=confirm)

    for _ in range(3):
        frame("now")
    comp.halo_shrink_after = 0          # (no probe frame inside the timed region)
    inexact_before = int(getattr(comp, "halo_inexact_frames", 0))
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        frame()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    if hasattr(comp, "confirm_halo"):
        comp.confirm_halo()             # (the last frame's verdict)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.one_device else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item()) / K * 1e3
    out = {"ms_per_frame": round(ms, 4), "Mpixels_per_s": round(fw * fh / ms / 1e3, 1), "frames": K,
           "exchange_confirmed": "two frames late, from pinned host memory behind an event (no drain in front of the composite, no host wait in the frame loop)",
           "exchange_inexact_frames": int(getattr(comp, "halo_inexact_frames", 0)) - inexact_before,
           "workload": f"opaque -> exchange -> mip chain -> transmissive -> composite, frame {fw}x{fh} in {world} row bands of {rows} rows, "
                       f"synthetic TGB-v1 layer as both layers, sun + {args.lights} punctual light(s)",
           "exchange": exchange, "exchange_fallbacks": int(getattr(comp, "halo_fallbacks", 0)),
           "composite": "RGBA16F frame, all-gather of the row bands"}
    del g, pyr, hdr
    if rank == 0 and not args.no_single_gpu_reference:
        gw = make_gbuffer_torch(fw, fh, dev)
        pw = OpaquePyramid(fw, fh, dev)
        hw = torch.zeros((fh, fw, 4), dtype=torch.float16, device=dev)
        for _ in range(3):
            r.record(gw, gw, uniforms, push, hw, pw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            r.record(gw, gw, uniforms, push, hw, pw)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / K * 1e3
        out["single_gpu_ms"] = round(one, 4)
        out["speedup_vs_1gpu"] = round(one / ms, 3)
        del gw, pw, hw
    dist.barrier()
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------ N = 1
def run_single(args) -> int:
    """The default command: the headline metric on one GPU and everything reported beside it (see the module docstring)."""
    import numpy as np
    import torch
    from transmission_renderer_amd import synthetic
    K, W = args.steps, args.warmup
    fw, fh = args.width, args.height
    wl = PassWorkload(0, fw, fh, lights=args.lights, roughness=args.roughness_override, all_transmissive=args.all_transmissive,
                      sets=args.sets, split=args.split)
    pixels = wl.pixels
    multi = len(wl.parts) > 1
    wl.ramp(args.ramp_s if args.ramp_s is not None else (0.15 if multi else 0.05))
    clock_ramp_launches = wl.launches
    if W > 0:
        wl.timed(W)                                               # W untimed warm-up steps
    # the metric: exactly K steps, wall clock between two device synchronisations
    ms_per_step, events_ms = wl.timed(K, first=W)
    value = pixels / ms_per_step / 1e3
    launch_log = [("clock_ramp", clock_ramp_launches), ("warmup", W * len(wl.parts)), ("timed", K * len(wl.parts))]
    # beside it: the same K steps over ONE input set (round 3's step), and one whole-frame launch per frame on one stream
    same_ms, _ = wl.timed(K, rotate=False)
    single_ms, _ = wl.timed(K, fn=wl.whole_frame)
    launch_log += [("same_input", K * len(wl.parts)), ("single_stream", K)]
    # p50 frame ms of the metric: the step's own pattern, timed in batches of 10 steps (an event pair per step would put a
    # barrier packet between the steps it measures)
    batch = 10     # (>= 50 timed steps whatever --steps says: SURVEY 8d)
    per_step = np.array([wl.timed(batch)[0] for _ in range(max(5, min(20, K // 2)))], dtype=np.float64)
    launch_log.append(("percentiles", len(per_step) * batch * len(wl.parts)))
    # ... and of single whole-frame launches, an event pair around each
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    for i in range(K):
        starts[i].record()
        wl.whole_frame(i)
        ends[i].record()
    torch.cuda.synchronize()
    per_launch_ms = np.array([a.elapsed_time(b) for a, b in zip(starts, ends)], dtype=np.float64)
    sync_ms = []
    for i in range(30):     # a launch + synchronise per step: what a host loop that waits for every frame sees
        t1 = time.perf_counter()
        wl.whole_frame(i)
        torch.cuda.synchronize()
        sync_ms.append((time.perf_counter() - t1) * 1e3)
    launch_log += [("per_launch_events", K), ("launch_sync", 30)]
    levels, n_sets, set_mb, lps = wl.levels, len(wl.sets), wl.set_bytes / 1e6, len(wl.parts)
    fig = wl.figures
    headline, same_fig, single_fig, ev_fig = fig(ms_per_step), fig(same_ms), fig(single_ms), fig(events_ms)
    scene_for_cpu, lut = wl.scene, wl.lut
    wl.close()
    torch.cuda.empty_cache()

    extras = not args.no_variants and not args.all_transmissive and args.roughness_override is None
    configs = {}
    if extras:
        kc, wc = min(K, 20), min(W, 5)
        try:
            configs["config2_1080p"] = measure_config("config2", 0, 1920, 1080, kc, wc, lights=1, with_record=True)
            configs["config3_4k"] = measure_config("config3", 0, 3840, 2160, kc, wc, lights=4, roughness=0.25)
            configs["config5_8k_1gpu"] = measure_config("config5", 0, 7680, 4320, kc, wc, lights=1)
            configs["all_transmissive"] = measure_config("all_transmissive", 0, fw, fh, kc, wc, lights=args.lights, all_transmissive=True)
        except Exception as e:       # (never costs the metric its line)
            configs["error"] = f"{type(e).__name__}: {e}"
    frame_pipeline = None
    if extras:
        try:
            frame_pipeline = frame_pipeline_time(fw, fh)
            frame_pipeline["at_1080p"] = {k: v for k, v in frame_pipeline_time(1920, 1080).items() if k in ("us_per_frame", "two_frames_in_flight")}
            frame_pipeline["at_8k"] = {k: v for k, v in frame_pipeline_time(7680, 4320).items() if k in ("us_per_frame", "two_frames_in_flight")}
            # the same frame where the rasteriser is NOT in its best case: the objects inside a closed room with partitions
            # behind them (the reference renders its model inside Sponza, src/main.rs:342-351; its depth pre-pass exists for
            # exactly this, readme.md:74) — every pixel covered, later draws nearer
            over = frame_pipeline_time(fw, fh, room=True)
            frame_pipeline["overdraw"] = {"us_per_frame": over["us_per_frame"], "two_frames_in_flight": over["two_frames_in_flight"],
                                          "at_1080p_us_per_frame": frame_pipeline_time(1920, 1080, room=True)["us_per_frame"],
                                          "fragments_per_pixel": round(fragments_per_pixel(True), 2),
                                          "fragments_per_pixel_of_the_plain_scene": round(fragments_per_pixel(False), 2),
                                          "scene": "the `meshes` scene inside a closed room (mesh_scene_parts(room=True)): back wall, "
                                                   "ceiling, side walls and three partitions drawn far to near behind the objects"}
            if not args.no_traffic:
                torch.cuda.synchronize()
                frame_pipeline["kernels"] = measure_frame_kernels(fw, fh)   # child processes; the GPU is idle here
                frame_pipeline["overdraw"]["kernels"] = measure_frame_kernels(fw, fh, room=True, counters=False)   # (one kernel-trace pass)
        except Exception as e:
            frame_pipeline = dict(frame_pipeline or {}, error=f"{type(e).__name__}: {e}")

    pmc = None if args.no_traffic else measure_pmc(args)          # child processes; the GPU is idle here
    ceiling = None if args.no_traffic else measure_pattern_ceiling()   # (a child process too)
    achieved = pixels * ALGORITHMIC_BYTES_PER_PIXEL / (ms_per_step * 1e-3) / 1e9
    out = {
        "metric": "shaded Mpixels/sec, 4K transmissive pass (fragment_transmission over a synthetic TGB-v1 G-buffer)",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round(ms_per_step, 4),
        "p50_frame_ms": round(float(np.percentile(per_step, 50)), 4), "p10_frame_ms": round(float(np.percentile(per_step, 10)), 4),
        "p90_frame_ms": round(float(np.percentile(per_step, 90)), 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"transmissive pass, frame {fw}x{fh}, TGB-v1 synthetic G-buffer fully covered, 16 materials"
                               + (" (all transmission_factor 1)" if args.all_transmissive else "")
                               + (f", roughness override {args.roughness_override}" if args.roughness_override is not None else "")
                               + f", sun + {args.lights} punctual light(s) (DragonAttenuation rig), RGBA16F target, "
                                 f"{levels}-level opaque pyramid, ggx_lut.png"
                               + (f", every frame shaded as {lps} row bands (one tr_shade_transmission call each) on {lps} HIP streams"
                                  if lps > 1 else "")
                               + f"; the timed steps rotate through {n_sets} distinct input sets of {set_mb:.0f} MB "
                                 "(planes + pyramid + target): no step re-reads what the previous one left in the caches",
                   "pixels_per_step": pixels, "pixels_per_gpu": pixels, "sharding": "none (one GPU holds the frame)",
                   "composite": "none (one GPU holds the frame)", "input_sets": n_sets, "input_set_MB": round(set_mb, 1)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "tr::shade_kernel<true, uint2, 0, false>", "avg_kernel_ms": round(ms_per_step, 4),
                     "algorithmic_bytes_per_pixel": ALGORITHMIC_BYTES_PER_PIXEL,
                     "algorithmic_bytes_per_launch": pixels * ALGORITHMIC_BYTES_PER_PIXEL,
                     "frac_survey_60B": headline["frac_60B"],
                     "frac_events": ev_fig["frac_52B"], "events_ms_per_step": round(events_ms, 4),
                     "streams": lps, "launches_per_step": lps,
                     "note": "frac = 52 B/px x pixels / ms_per_step (the wall clock of the K timed steps, the same number `value` "
                             "comes from) / 8 TB/s.  52 B/px = what this untextured variant moves (16 + 16 + 4 B planes, 8 B opaque "
                             "colour, 8 B write; = SURVEY 8d's read-only figure); SURVEY 8d's 60 B/px also counts the 8 B/px uv "
                             "plane, which this kernel never loads (frac_survey_60B).  frac_events: the same region bracketed by "
                             "HIP events on the launch streams.  A step is "
                             f"{lps} launch(es) of the kernel, one per row band of the frame, each on its own stream; `traffic` is "
                             "and algorithmic_bytes_per_launch are per WHOLE-FRAME launch (the --pmc-probe child runs)"},
        "same_input": dict(same_fig, note="the same K steps over ONE input set (round 3's timed step): what re-reading a G-buffer "
                                          "that the previous step left in the Infinity Cache is worth"),
        "single_stream": dict(single_fig, avg_kernel_ms=round(single_ms, 4), frac=single_fig["frac_52B"], p50_launch_ms=round(float(np.percentile(per_launch_ms, 50)), 4),
                              p10_launch_ms=round(float(np.percentile(per_launch_ms, 10)), 4),
                              p90_launch_ms=round(float(np.percentile(per_launch_ms, 90)), 4),
                              note="ONE whole-frame tr_shade_transmission call per frame on one stream, each behind the "
                                   "previous one, rotating input sets (rounds 1-2's step; what a caller that does not split the pass "
                                   "gets); p*_launch_ms: an event pair around every launch"),
        "clock_ramp_launches": clock_ramp_launches,
        "launch_sync_p50_ms": round(float(np.percentile(sync_ms, 50)), 4),
        "launch_log": launch_log,
    }
    if ceiling is not None:
        # the pass against what its own memory pattern attains with no arithmetic at all, same run, same launch shape
        disp, same = ceiling["pattern_taps_displaced_48px_us"], ceiling["pattern_taps_not_displaced_us"]
        frac_of = lambda us: round(pixels * ALGORITHMIC_BYTES_PER_PIXEL / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)   # noqa: E731
        out["roofline"]["pattern_ceiling"] = dict(
            ceiling, bytes_per_pixel=ALGORITHMIC_BYTES_PER_PIXEL,
            attainable_frac_of_peak={"taps_displaced_48px": frac_of(disp["two_bands"]), "taps_not_displaced": frac_of(same["two_bands"])},
            pass_time_over_pattern_time={"two_bands": [round(ms_per_step * 1e3 / disp["two_bands"], 3), round(ms_per_step * 1e3 / same["two_bands"], 3)],
                                         "one_call": [round(single_ms * 1e3 / disp["one_call"], 3), round(single_ms * 1e3 / same["one_call"], 3)]},
            note="tools/ubench/pattern_ceiling.hip --brief, run behind the timed region: an ARITHMETIC-FREE kernel issuing the pass's "
                 "requests (two non-temporal plane rows, the id row, four gathered 16-byte texel pairs per pixel — displaced from "
                 "the pixel by a smooth field of up to 48 px, or not at all: the scene's refraction lies between —, a LUT line, the "
                 "8-byte store) on the pass's tile numbering, 8 waves per SIMD, cold inputs, as two bands on two streams and as one "
                 "call per frame.  attainable_frac_of_peak = 52 B/px x pixels / its two-band time / 8 TB/s: what this access pattern "
                 "reaches on this box with nothing to compute; pass_time_over_pattern_time [vs displaced, vs not displaced] <= 1: "
                 "the pass, with all its arithmetic, is no slower than its own memory pattern")
    tr_ = None
    if pmc is not None:
        c = pmc["headline"]
        tr_ = {"hbm_bytes_per_launch": c["hbm_bytes_per_launch"], "source": pmc["source"]}
        out["roofline"]["valu"] = valu_roofline(c, pixels, ms_per_step)
        if "config3_4k" in configs and isinstance(configs["config3_4k"], dict):
            configs["config3_4k"]["valu"] = valu_roofline(pmc["config3"], 3840 * 2160, configs["config3_4k"]["us"] * 1e-3)
            configs["config3_4k"]["traffic"] = pmc["config3"]["hbm_bytes_per_launch"]
        if "all_transmissive" in configs and isinstance(configs["all_transmissive"], dict):
            configs["all_transmissive"]["valu"] = valu_roofline(pmc["all_transmissive"], pixels, configs["all_transmissive"]["us"] * 1e-3)
            configs["all_transmissive"]["traffic"] = pmc["all_transmissive"]["hbm_bytes_per_launch"]
    traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if tr_ is None and os.path.exists(traffic_file) and not args.all_transmissive and args.roughness_override is None:
        try:
            with open(traffic_file) as f:
                committed = json.load(f)
            if committed.get("width") == fw and committed.get("height") == fh and committed.get("lights") == args.lights:
                tr_ = committed
        except Exception:
            pass
    if tr_ is not None:
        kernel_s = ms_per_step * 1e-3
        out["roofline"]["traffic"] = tr_["hbm_bytes_per_launch"]
        out["roofline"]["achieved_from_traffic"] = round(tr_["hbm_bytes_per_launch"] / kernel_s / 1e9, 1)
        out["roofline"]["frac_from_traffic"] = round(tr_["hbm_bytes_per_launch"] / kernel_s / 1e9 / HBM_PEAK_GBS, 4)
        out["roofline"]["traffic_source"] = tr_.get("source")
    if configs:
        out["configs"] = configs
        if "all_transmissive" in configs and isinstance(configs["all_transmissive"], dict):     # (the key rounds 2-3 reported)
            a = configs["all_transmissive"]
            out["variants"] = {"all_transmissive": {"avg_kernel_ms": round(a["us"] / 1e3, 4), "Mpixels_per_s": a["Mpixels_per_s"],
                                                    "frac": a["frac_52B"], "frac_survey_60B": a["frac_60B"]}}
    if frame_pipeline:
        out["frame_pipeline"] = frame_pipeline
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene_for_cpu, lut, fw, fh, args.cpu_budget_s)
        out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 3)
    print(json.dumps(out), flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ one rank (N > 1)
def run_rank(args) -> int:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)
    if args.pmc_probe:
        return pmc_probe(args)
    if args.frame_probe:
        return frame_probe(args)
    if world == 1 and not args.rehearse_distributed and args.streams <= 1:
        return run_single(args)
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
        if args.one_device:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
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
    g = make_gbuffer_torch(fw, fh, dev, rows=(y0, y1) if y1 > y0 else (0, 1))   # this rank's screen tile only
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
    present_ldr = distributed and composite != "none" and args.composite_format in ("rgba8", "rgb8")
    ldr_channels = 3 if args.composite_format == "rgb8" else 4
    ldr_frames = [torch.zeros((padded, fw, ldr_channels), dtype=torch.uint8, device=dev) for _ in frames] if present_ldr else None
    tonemap_params = r.baked_tonemap_params() if present_ldr else None
    comp = sharded.Compositor(world, rank, renderer=r, single_rank_comm=args.rehearse_distributed) if distributed else None
    if distributed and args.require_rccl and not comp.backend.startswith("tr_allgather_frame"):
        print(f"bench.py --require-rccl: the composite would run over {comp.backend!r}, not over tr_allgather_frame (RCCL)",
              file=sys.stderr, flush=True)
        dist.barrier()
        dist.destroy_process_group()
        r.close()
        return 3
    cdev = "cpu" if args.one_device else dev     # where the run's small book-keeping collectives live (gloo reduces host memory)
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
            (r.tonemap_rgb8 if ldr_channels == 3 else r.tonemap)(frames[i][y0:y1], tonemap_params, out=ldr_frames[i][y0:y1])
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
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
        mine = torch.tensor([kernel_ms, float((y1 - y0) * fw)], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_kernel_ms = [float(e[0].item()) for e in every]
        per_rank_pixels = [float(e[1].item()) for e in every]
        kernel_ms_max = max(per_rank_kernel_ms)
        if kernel_in_flight_ms is not None:
            t = torch.tensor([kernel_in_flight_ms], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            kernel_in_flight_ms = float(t.item())
        dist.barrier()
        composite_ms = timed_launches(max(10, K // 4), lambda: comp.allgather_rows(frames[0]))
        t = torch.tensor([composite_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        composite_ms = float(t.item())
        # the same composite of the frame as it is presented: bands tonemapped to RGBA8 first (half the bytes per link)
        ldr = torch.zeros((padded, fw, ldr_channels), dtype=torch.uint8, device=dev)
        dist.barrier()
        composite_ldr_ms = timed_launches(max(10, K // 4), lambda: comp.allgather_rows(ldr))
        t = torch.tensor([composite_ldr_ms], dtype=torch.float64, device=cdev)
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
            gw = make_gbuffer_torch(fw, fh, dev)
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

    # BASELINE config 5's frame through the sharded FULL pipeline (opaque band -> exchange of the opaque colour -> mip chain
    # -> transmissive band -> composite): the workload where sharding can pay — per rank 1/N of two shading passes against one
    # exchange, where the transmissive pass alone is ~1/N of 360 us against a composite that costs as much.
    full_pipeline = None
    if distributed and strong and not args.no_full_pipeline:
        full_pipeline = full_pipeline_8k(r, comp, world, rank, dev, dist, scene, max(5, min(K, 20)), args)

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
                                        ", of the frame as presented: every rank tonemaps its band (tr_tonemap" + ("_rgb8" if ldr_channels == 3 else "") + ") inside the step and "
                                        f"the {'RGB8' if ldr_channels == 3 else 'RGBA8'} bands ({ldr_channels} B/px) are gathered" if present_ldr else
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
            out["composite_backend"] = comp.backend
            out["rccl_ranks"] = comp.rccl_ranks          # ncclCommCount of the library's communicator; null: it is not in use
            out["composite_fell_back"] = bool(comp.fell_back)
            if args.one_device:
                out["one_device_rehearsal"] = ("every rank on cuda:0, gloo process group, exchanges staged through host memory: the "
                                               "N > 1 code path on device buffers, NOT a measurement of the links")
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
            out["composite_rgba8_allgather_ms" if ldr_channels == 4 else "composite_rgb8_allgather_ms"] = round(composite_ldr_ms, 4)
            if single_gpu_ms is not None:
                out["single_gpu_ms"] = round(single_gpu_ms, 4)
                out["speedup_vs_1gpu"] = {"kernel_only": round(single_gpu_ms / kernel_ms_max, 3),
                                          "with_composite": round(single_gpu_ms / ms_per_step, 3)}
                if kernel_in_flight_ms is not None:   # (both sides with two frames in flight)
                    out["speedup_vs_1gpu"]["kernel_only_two_frames_in_flight"] = round(single_gpu_in_flight_ms / kernel_in_flight_ms, 3)
        if full_pipeline:
            out["full_pipeline_8k"] = full_pipeline
            out["values"] = {"value": "transmissive pass of the 3840x2160 frame in row bands + composite (BASELINE config 4: the metric)",
                             "full_pipeline_8k.Mpixels_per_s": "opaque -> exchange -> mips -> transmissive -> composite of the 7680x4320 frame "
                                                               "(BASELINE config 5), all ranks together"}
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
