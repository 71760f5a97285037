"""`python bench.py --gpus N` starts by itself (no torchrun needed): the parent spawns N ranks with the rendezvous
environment set, the ranks cut the frame into tr_band_rows bands and composite it.  `--selftest-cpu` runs exactly that
launcher and band arithmetic over gloo on host tensors, so it is covered here without a GPU; the GPU body of the
ranks runs on the GPU box (driver: BENCH / SCALE)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                       text=True, timeout=280)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n,height", [(2, 54), (3, 50)])
def test_bench_self_launches_n_ranks(n, height):
    p, out = _run("--gpus", str(n), "--selftest-cpu", "--height", str(height))
    assert p.returncode == 0, p.stderr[-2000:]
    assert out and out["selftest"] == "ok" and out["n_ranks"] == n
    rows = out["rows_per_rank"]
    assert rows % 4 == 0 and rows * n >= height
    assert out["bands"][0][0] == 0 and out["bands"][-1][1] == height
    assert all(out["bands"][i][1] == out["bands"][i + 1][0] for i in range(n - 1))
    assert out["composite"] == "torch.distributed:gloo"
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1      # ONE JSON line, from rank 0


@pytest.mark.timeout(300)
def test_bench_under_an_external_launcher_uses_its_environment():
    """torch.distributed.run style: WORLD_SIZE is set, so bench.py is a rank, not a launcher."""
    p, out = _run("--gpus", "1", "--selftest-cpu", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 0 and out["n_ranks"] == 1 and out["bands"] == [[0, 54]]


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launch():
    p, out = _run("--gpus", "2", "--selftest-cpu", "--height", "0")     # tr_band_rows refuses an empty frame
    assert p.returncode != 0 and out is None
