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


_EMITTED = {}


@pytest.mark.gpu
@pytest.mark.parametrize("via_mixin", [False, True])
def test_plain_python_bench_gpus2_on_one_gpu(via_mixin):
    """The exact command of the driver's scaling run, two ranks sharing cuda:0 over gloo (TDC_BENCH_ONE_GPU / TDC_DIST_BACKEND
    are the test hooks of bench.py): one JSON line with the per-rank times.  via_mixin: the same through the drop-in boundary -
    every rank calls prepare_inputs_labels_for_multimodal with the whole video and config.tdc_shard_frames splits its frames
    over the ranks (model.py -> dist.ShardedVideoEncoder); the emitted stream has the same length either way."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "64", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline"] + (["--via-mixin"] if via_mixin else []),
                       env=_env(TDC_BENCH_ONE_GPU="1", TDC_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 1 and res["value"] > 0
    rk = res["rank_ms_per_step"]
    assert len(rk["per_rank"]) == 2 and rk["max"] >= rk["min"] > 0
    assert res["roofline"]["frac"] > 0 and res["config"]["frames"] == 64
    assert res["ranks_seen"] == [0, 1] and res["rank_devices"] == [0, 0] and res["dist_backend"] == "gloo"
    assert res["config"]["entry"].startswith("mixin" if via_mixin else "engine")
    assert res["config"]["product_setting"]["tdc_frame_cap"] == 64
    _EMITTED[via_mixin] = res["config"]["emitted_tokens"]
    if len(_EMITTED) == 2:
        assert _EMITTED[False] == _EMITTED[True] > 64 * 17


def test_launch_guard_stops_a_rank_that_never_finishes():
    """One rank finishes, the other sleeps for ever (a peer stuck in a collective): the launcher must come back within its
    wall-clock guard with code 124, having stopped the sleeper - not hold the caller's lease."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    child = [sys.executable, "-c", "import os, time\nif os.environ['RANK'] == '1':\n    time.sleep(10 ** 6)\n"]
    t0 = time.monotonic()
    rc = bench.self_launch(2, guard_s=3.0, cmd=child)
    dt = time.monotonic() - t0
    assert rc == 124
    assert dt < 40, dt


def test_launch_without_stragglers_is_unaffected_by_the_guard():
    sys.path.insert(0, ROOT)
    import bench
    ok = [sys.executable, "-c", "import os; assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'"]
    assert bench.self_launch(2, guard_s=60.0, cmd=ok) == 0
    bad = [sys.executable, "-c", "import os, sys, time\nif os.environ['RANK'] == '0':\n    sys.exit(7)\ntime.sleep(10 ** 6)\n"]
    assert bench.self_launch(2, guard_s=60.0, cmd=bad) == 7


def test_rank_guard_ends_a_stuck_rank_with_124():
    code = ("import sys, time; sys.path.insert(0, %r); import bench; bench.rank_guard(1.0); time.sleep(60)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 124 and "rank-timeout" in r.stderr


@pytest.mark.gpu
def test_torchrun_form_of_the_drivers_scaling_run_on_one_gpu():
    """The driver's N > 1 command line - python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ... - with both ranks on cuda:0 over gloo (the bench's test hooks): bench.py must run as a
    rank (not launch ranks of its own), print ONE JSON line on rank 0 and leave the process group cleanly."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "64",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=_env(TDC_BENCH_ONE_GPU="1", TDC_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == [0, 1] and res["value"] > 0 and res["scaling"] == "strong"
