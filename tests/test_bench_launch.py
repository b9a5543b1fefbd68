"""bench.py --gpus N started WITHOUT a launcher (the driver's command form) must bring up its own N ranks as fresh child
processes.  On a box without a GPU every rank gets as far as the "no GPU" exit: the parent reports the ranks' exit codes,
prints no result line and fails — it neither hangs nor falls back to anything."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="the dry start is a CPU-box check; on a GPU box the driver runs the real thing")
@pytest.mark.parametrize("n", [2, 8])
def test_bench_starts_its_own_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2
    assert r.stderr.count("no GPU visible") == n                 # every rank started and reached the device check
    assert f"rank exit codes {[2] * n}" in r.stderr
    assert r.stdout.strip() == ""                                # no result line without a result


def test_bench_refuses_a_mismatched_launcher():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
