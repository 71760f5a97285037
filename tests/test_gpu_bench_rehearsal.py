"""bench.py's N > 1 code path rehearsed on the one GPU of the test box (run with -m gpu): a process group of one rank,
the library's own RCCL communicator (tr_comm_create / tr_allgather_frame), the composite pipelined on a second stream
inside the timed step, the per-rank gathers and the single-GPU reference — every line of the multi-GPU bench that one
GPU can execute.  (N ranks cannot share a GPU under RCCL; the launcher and the band arithmetic are covered on CPU.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("fmt", ["rgba8", "rgba16f"])
def test_distributed_bench_path_on_one_gpu(fmt):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-distributed", "--steps", "20", "--warmup", "3",
                        "--width", "1920", "--height", "1080", "--no-cpu-baseline", "--composite-format", fmt], env=env, capture_output=True, text=True, timeout=580)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["composite"].startswith("overlap: tr_allgather_frame (RCCL), of the "
                                                 + ("frame as presented" if fmt == "rgba8" else "RGBA16F HDR target"))
    assert out["kernel_only"]["per_rank_kernel_ms"][0] > 0 and 0 < out["kernel_only"]["per_rank_roofline_frac"][0] < 1
    assert out["composite_allgather_ms"] > 0 and out["composite_rgba8_allgather_ms"] > 0
    assert out["single_gpu_ms"] > 0 and out["speedup_vs_1gpu"]["kernel_only"] > 0.5
    assert out["kernel_only"]["two_frames_in_flight"]["ms_per_step"] > 0 and out["speedup_vs_1gpu"]["kernel_only_two_frames_in_flight"] > 0.5
