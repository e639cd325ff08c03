"""Multi-GPU readiness (pytest -m gpu): exercised on RCCL whenever the box has >= 2 HIP devices, skipped otherwise -- so the day the
driver runs the suite on a multi-GPU node the N-rank paths (bench.py --gpus 2, gradient all-reduce on shared rows, TSDF fusion) are
tested without anyone editing a file.  On the 1-GPU box only the launcher's pinning logic runs."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
need2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 HIP devices")


def _torchrun(script_args, nproc=2, port=29655, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


@need2
def test_bench_two_ranks_over_rccl():
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-method-iteration"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 5


@need2
def test_collectives_two_ranks_over_rccl():
    r = _torchrun([os.path.join(ROOT, "tests", "multi_rank_worker.py")], port=29656)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "multi_rank_worker ok" in r.stdout


def test_launcher_pins_workers(tmp_path):
    """One worker per GPU with HIP_VISIBLE_DEVICES set per child (runs with --gpus 1 x 2 children sharing the single GPU when only one
    device exists: the pinning and the failure propagation of spawn() are what is under test here, not the collectives)."""
    sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd"))
    from gsrast import launch_tiles
    data = tmp_path / "scene"
    for i in range(2):
        (data / f"tile_{i:04d}").mkdir(parents=True)
    n = min(2, max(1, torch.cuda.device_count()))
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.path.join(ROOT, "gs-sr_amd"))
    r = subprocess.run([sys.executable, "-m", "gsrast.launch_tiles", "--data", str(data), "--output", str(tmp_path / "out"), "--entry",
                        "multi_rank_worker:tile_entry", "--gpus", str(n), "--backend", "nccl", "--port", "29657"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    summ = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert summ["tiles"] == 2 and summ["iterations"] == 2 * 7
    for i in range(2):
        rec = json.load(open(tmp_path / "out" / f"tile_{i:04d}" / "config" / "worker.json"))
        assert rec["visible"] is not None and len(rec["visible"].split(",")) == 1 and rec["device_count"] == 1
    # a failing rank ends the job instead of hanging it
    r = subprocess.run([sys.executable, "-m", "gsrast.launch_tiles", "--data", str(data), "--output", str(tmp_path / "out2"), "--entry",
                        "multi_rank_worker:failing_entry", "--gpus", str(max(n, 1)), "--backend", "nccl", "--port", "29658"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert launch_tiles.gpu_numa_cpus(0) is None or len(launch_tiles.gpu_numa_cpus(0)) > 0
