"""Multi-rank paths (pytest -m gpu).

Run on every box, also with ONE device: `bench.py --gpus 2 --oversubscribe` launching its own two ranks, and the tile launcher with two
workers on one GPU -- both go through gsrast.launch_tiles.spawn_ranks (per-child HIP_VISIBLE_DEVICES, NUMA affinity, polling, failure
propagation) with real control collectives (gloo on host tensors when ranks share a device, RCCL otherwise).
Run only with >= 2 HIP devices (skipped otherwise): the same over RCCL under torch.distributed.run, the gradient all-reduce on shared
rows and the cross-rank TSDF fusion."""
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


def _bench(extra, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GSR_BENCH_BACKEND"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun must be a 2-rank job.  On a 1-GPU box the two ranks share the device
    (--oversubscribe; control collectives over gloo), with >= 2 devices each rank has its own and the backend is RCCL."""
    nd = torch.cuda.device_count()
    r = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--P", "60000", "--W", "960", "--H", "544", "--oversubscribe",
                "--no-cpu-baseline", "--no-method-iteration"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist_world_size"] == 2 and d["steps"] == 5 and d["value"] > 0 and d["scaling"] == "weak"
    assert [q["rank"] for q in d["ranks"]] == [0, 1] and len({q["pid"] for q in d["ranks"]}) == 2
    assert all(q["visible"] is not None and len(q["visible"].split(",")) == 1 for q in d["ranks"])          # every rank pinned to one device
    assert d["backend"] == ("nccl" if nd >= 2 else "gloo") and d["distinct_devices"] == min(2, nd)


def test_bench_eight_ranks_on_whatever_devices_there_are():
    """The command the driver runs on an 8-GPU node, `bench.py --gpus 8`, at reduced size on THIS box: eight pinned processes, eight tile-scenes (seed = rank),
    one JSON line with the whole-job rate.  With fewer than 8 devices the ranks share them (--oversubscribe, control collectives over gloo): every rank still
    runs the product's forward + backward + Adam on its own scene."""
    nd = torch.cuda.device_count()
    r = _bench(["--gpus", "8", "--steps", "3", "--warmup", "1", "--P", "20000", "--W", "640", "--H", "368", "--no-cpu-baseline", "--no-method-iteration"] +
               (["--oversubscribe"] if nd < 8 else []), timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["dist_world_size"] == 8 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak" and not d.get("dry_run")
    assert [q["rank"] for q in d["ranks"]] == list(range(8)) and len({q["pid"] for q in d["ranks"]}) == 8
    assert all(q["visible"] is not None and len(q["visible"].split(",")) == 1 for q in d["ranks"])
    assert d["backend"] == ("nccl" if nd >= 8 else "gloo") and d["distinct_devices"] == min(8, nd)


def test_bench_refuses_more_ranks_than_devices():
    nd = torch.cuda.device_count()
    r = _bench(["--gpus", str(nd + 1), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-method-iteration"], timeout=120)
    assert r.returncode != 0 and "--oversubscribe" in (r.stderr + r.stdout)
    # a rank count that does not match --gpus is refused as well (the round-2 failure mode: --gpus parsed and ignored)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29659")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_launcher_pins_workers(tmp_path):
    """spawn() on the GPU box: one worker per visible GPU, or -- with a single device -- two workers on it (--workers-per-gpu 2), so the
    children, their HIP_VISIBLE_DEVICES pinning, the polling and the real control collectives (barrier + job reduction; over gloo when the
    ranks share the device, ADVICE r2: RCCL rejects duplicate devices) all run even on one GPU."""
    sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd"))
    from gsrast import launch_tiles
    data = tmp_path / "scene"
    for i in range(2):
        (data / f"tile_{i:04d}").mkdir(parents=True)
    nd = torch.cuda.device_count()
    shape = ["--gpus", "2"] if nd >= 2 else ["--gpus", "1", "--workers-per-gpu", "2"]
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.path.join(ROOT, "gs-sr_amd"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "gsrast.launch_tiles", "--data", str(data), "--output", str(tmp_path / "out"), "--entry",
                        "multi_rank_worker:tile_entry", "--backend", "nccl", "--port", "29657"] + shape,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    summ = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert summ["tiles"] == 2 and summ["workers"] == 2 and summ["iterations"] == 2 * 7 and summ["device"] == "cuda"
    assert summ["backend"] == ("nccl" if nd >= 2 else "gloo")
    pids = set()
    for i in range(2):
        rec = json.load(open(tmp_path / "out" / f"tile_{i:04d}" / "config" / "worker.json"))
        assert rec["visible"] is not None and len(rec["visible"].split(",")) == 1 and rec["device_count"] == 1 and rec["device"].startswith("cuda")
        pids.add(rec["pid"])
    assert len(pids) == 2                                   # two child processes really ran (not the in-process shortcut)
    # a failing rank ends the job instead of hanging it
    r = subprocess.run([sys.executable, "-m", "gsrast.launch_tiles", "--data", str(data), "--output", str(tmp_path / "out2"), "--entry",
                        "multi_rank_worker:failing_entry", "--backend", "nccl", "--port", "29658"] + shape,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert launch_tiles.gpu_numa_cpus(0) is None or len(launch_tiles.gpu_numa_cpus(0)) > 0
    # the KFD-derived PCI address of every visible device is the one the HIP runtime reports for that ordinal
    visible = os.environ.get("HIP_VISIBLE_DEVICES")
    ids = [v for v in visible.split(",") if v] if visible else [str(i) for i in range(torch.cuda.device_count())]
    bdfs = launch_tiles.hip_device_bdfs()
    if not os.environ.get("ROCR_VISIBLE_DEVICES") and all(i.isdigit() for i in ids):
        for k, i in enumerate(ids[:torch.cuda.device_count()]):
            pr = torch.cuda.get_device_properties(k)
            assert bdfs[int(i)][0] == f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0", (bdfs, k, pr.pci_bus_id)
