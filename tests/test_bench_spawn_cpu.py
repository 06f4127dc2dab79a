"""bench.py --gpus N outside a torchrun environment must start N ranks itself (a plain
`python3 bench.py --gpus 8` used to run one process and print n_gpus 1).  Dry run: rank plumbing only, gloo, no GPU.
Reference launch being stood in for: scripts/train.py:1044-1049 (mp.spawn of train_ddp, one process per GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                       timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_gpus_2_spawns_two_ranks():
    out = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "3", "--warmup", "1")
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 64
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["max_rank_elapsed_s"] >= 0.02  # MAX over ranks: rank 1 sleeps twice as long as rank 0


@pytest.mark.timeout(120)
def test_gpus_1_stays_in_process():
    out = _run("--dry-run")
    assert out["n_gpus"] == 1
