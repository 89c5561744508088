"""bench.py as the driver starts it.  `python bench.py --gpus N` (N > 1, no RANK in the environment) must start its own N ranks
(one per GPU, the reference's launch convention: /root/reference/eval/eval_mlvu.py:129-157), print rank 0's ONE JSON line and
return the worst child's code; under torch.distributed.run it must keep working as a rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_self_launch_propagates_a_failing_rank_without_hanging():
    """No GPU here: every child dies at torch.cuda.set_device - the launcher must come back with a non-zero code, print no
    JSON line, and never touch the GPU itself (it would raise in this process otherwise)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "8", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "stopping the other ranks" in r.stderr or "exited with code" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error_message_not_an_assert():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"],
                       env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "Traceback" not in r.stderr


@pytest.mark.gpu
def test_plain_python_bench_gpus2_on_one_gpu():
    """The exact command of the driver's scaling run, two ranks sharing cuda:0 over gloo (TDC_BENCH_ONE_GPU / TDC_DIST_BACKEND
    are the test hooks of bench.py): one JSON line with the per-rank times."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "64", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline"], env=_env(TDC_BENCH_ONE_GPU="1", TDC_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 1 and res["value"] > 0
    rk = res["rank_ms_per_step"]
    assert len(rk["per_rank"]) == 2 and rk["max"] >= rk["min"] > 0
    assert res["roofline"]["frac"] > 0 and res["config"]["frames"] == 64
