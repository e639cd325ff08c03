"""Host logic of bench.py that the judged JSON line rests on (no GPU): the algorithmic-byte model of SURVEY §8(d), the mirror of the library's
depth-order rule, the self-launcher's environment, and the argument defaults the driver relies on."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402


def test_algorithmic_bytes_follow_the_survey_table():
    # bytes staged per tile instance, read / written per pixel, gradient bytes per instance; + 8 B per tile range
    R, N, T = 1_380_000, 1920 * 1080, 8160
    for variant, (rec, pix_f, grad, pix_b) in {"ewa": (40, 20, 44, 20), "plane": (60, 44, 68, 64), "surfel": (76, 76, 72, 76)}.items():
        fwd, bwd = bench.algorithmic_bytes(variant, R, N, T)
        assert fwd == R * rec + N * pix_f + 8 * T
        assert bwd == R * (rec + grad) + N * pix_b + 8 * T
    # the headline kernel's figure in the committed bench line is this model at the R that run measured
    import json
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_default.json")).read().strip().splitlines()[-1])
    fwd, bwd = bench.algorithmic_bytes("surfel", line["config"]["tile_instances_R"], N, T)
    assert line["roofline"]["algorithmic_bytes_per_launch"] == bwd
    assert abs(line["roofline"]["achieved"] - bwd / (line["roofline"]["avg_launch_ms"] * 1e-3) / 1e9) < 0.5
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4


def test_depth_order_rule_mirrors_the_library(monkeypatch):
    src = open(os.path.join(ROOT, "gs-sr_amd", "csrc", "gsr_binning.hip")).read()
    m = re.search(r"variant == GSR_EWA \? (\d+)ll : \(variant == GSR_PLANE \? (\d+)ll : (\d+)ll\)", src)
    assert m, "gsr_depth_order_static_rule changed shape: update bench.depth_order_is_global and this test"
    per_tile = {"ewa": int(m.group(1)), "plane": int(m.group(2)), "surfel": int(m.group(3))}
    monkeypatch.delenv("GSR_DEPTH_ORDER", raising=False)
    T = 8160
    for v, k in per_tile.items():
        assert bench.depth_order_is_global(k * T, T, v) is False          # P <= k T: per-tile sort inside the forward
        assert bench.depth_order_is_global(k * T + 1, T, v) is True
    monkeypatch.setenv("GSR_DEPTH_ORDER", "global")
    assert bench.depth_order_is_global(1, T, "surfel") is True
    monkeypatch.setenv("GSR_DEPTH_ORDER", "tile")
    assert bench.depth_order_is_global(10 ** 9, T, "ewa") is False


def test_stage_bytes_move_the_sort_into_the_forward_when_it_is_fused(monkeypatch):
    monkeypatch.delenv("GSR_DEPTH_ORDER", raising=False)
    P, R, N, T = 300_000, 1_380_000, 1920 * 1080, 8160
    b = bench.stage_bytes("surfel", "precomp", P, R, N, T)
    fwd, bwd = bench.algorithmic_bytes("surfel", R, N, T)
    assert b["blend_bwd"] == bwd
    assert b["blend_fwd"] == fwd + 12 * R          # per-tile mode: the prologue reads id + depth key and writes the id back
    assert set(b) >= {"preprocess", "depth_order", "binning", "blend_fwd", "blend_bwd", "preprocess_bwd"}


def test_helper_children_do_not_inherit_the_rendezvous(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.setenv(k, "1")
    monkeypatch.setenv("GSR_XCD_REMAP", "2")
    env = bench.child_env()
    assert not {"RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"} & set(env)
    assert env["GSR_XCD_REMAP"] == "2"


def test_self_launcher_refuses_shared_devices_unless_asked(monkeypatch):
    import types
    import pytest
    from gsrast import launch_tiles
    seen = {}

    def fake_spawn(cmd_of_rank, world, ndev, port, pin_gpus=True, extra_env=None):
        seen.update(world=world, ndev=ndev, port=port, pin=pin_gpus, env=dict(extra_env or {}), cmd=cmd_of_rank(0))
        return 0
    monkeypatch.setattr(launch_tiles, "spawn_ranks", fake_spawn)
    monkeypatch.setenv("MASTER_PORT", "29577")
    args = types.SimpleNamespace(gpus=4, oversubscribe=False, no_pin=False)
    with pytest.raises(SystemExit):
        bench.launch_ranks(args, ndev=2)                       # two ranks on one device: RCCL would refuse the communicator
    assert bench.launch_ranks(types.SimpleNamespace(gpus=2, oversubscribe=False, no_pin=False), ndev=8) == 0
    assert seen["world"] == 2 and seen["ndev"] == 2 and seen["port"] == 29577 and seen["pin"] and seen["env"] == {"GSR_BENCH_BACKEND": "nccl"}
    assert os.path.basename(seen["cmd"][1]) == "bench.py"
    assert bench.launch_ranks(types.SimpleNamespace(gpus=4, oversubscribe=True, no_pin=False), ndev=2) == 0
    assert seen["world"] == 4 and seen["ndev"] == 2 and seen["env"] == {"GSR_BENCH_BACKEND": "gloo"}


def _dry(cmd, env_extra=None, timeout=180):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GSR_BENCH_BACKEND", "HIP_VISIBLE_DEVICES")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable] + cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_eight_rank_launch_path_without_a_device():
    """no 8-GPU node has ever run this repo.  `bench.py --gpus 8 --dry-run` walks the whole launch path of the 8-GPU command on CPU -- eight
    processes with the torchrun environment contract, rank r pinned to device r (HIP_VISIBLE_DEVICES), one gloo group of 8, barrier-bracketed timed region,
    max / sum reduction, exactly ONE JSON line from rank 0 with n_gpus 8 -- and its line says what it is (value null, 'DRY RUN').  Both ways of starting it:
    self-launched, and under torch.distributed.run as the driver does for N > 1."""
    import json
    import sys
    for how in ("self", "torchrun"):
        if how == "self":
            r = _dry([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "4", "--warmup", "1"], {"MASTER_PORT": "29731"})
        else:
            r = _dry(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29733",
                      os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "4", "--warmup", "1"])
        assert r.returncode == 0, (how, r.stdout[-1500:], r.stderr[-2500:])
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, (how, r.stdout[-2000:])
        d = json.loads(lines[0])
        assert d["n_gpus"] == 8 and d["dist_world_size"] == 8 and d["steps"] == 4 and d["total_steps_over_ranks"] == 32 and d["scaling"] == "weak"
        assert d["dry_run"] is True and d["value"] is None and "DRY RUN" in d["metric"]
        assert [q["rank"] for q in d["ranks"]] == list(range(8)) and len({q["pid"] for q in d["ranks"]}) == 8
        if how == "self":          # the self-launcher pins: rank r sees exactly device r, as LOCAL_RANK 0 of its own one-device world
            assert [q["visible"] for q in d["ranks"]] == [str(i) for i in range(8)] and all(q["local_rank"] == "0" for q in d["ranks"])
        else:                      # torchrun does not pin: LOCAL_RANK selects the device inside bench.py
            assert [q["local_rank"] for q in d["ranks"]] == [str(i) for i in range(8)]
    # a rank count that does not match --gpus is refused in the dry run exactly as in a real one
    r = _dry([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], {"RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29735"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_committed_rocprof_summary_agrees_with_the_bench_line():
    """profiles/: the rocprofv3 --kernel-trace --stats average of the dominant kernel and the HIP-event average inside bench.py are two
    measurements of the same launches (different boxes of the pool: a few per cent apart at most), and the PMC traffic figure quoted in the
    line is the one in the committed counter summary."""
    import csv
    import json
    prof = os.path.join(ROOT, "profiles")
    line = json.loads(open(os.path.join(prof, "r06_bench_default.json")).read().strip().splitlines()[-1])
    rows = [r for r in csv.DictReader(open(os.path.join(prof, "r06_surfel_kernel_stats.csv"))) if "k_blend_bwd_sp<1>" in r["Name"]]
    assert len(rows) == 1
    rocprof_ms = float(rows[0]["AverageNs"]) * 1e-6
    assert abs(rocprof_ms - line["roofline"]["avg_launch_ms"]) < 0.03 * line["roofline"]["avg_launch_ms"]
    pmc = json.load(open(os.path.join(prof, "r06_pmc_summary.json")))
    assert abs(pmc["k_blend_bwd_sp<1>"]["hbm_bytes_per_launch"] - line["roofline"]["traffic"]) < 0.02 * line["roofline"]["traffic"]
    assert line["roofline"]["traffic"] >= line["roofline"]["algorithmic_bytes_per_launch"]
    assert line["metric"].startswith("train iters/sec") and line["unit"] == "iters/s" and line["n_gpus"] == 1 and line["scaling"] == "weak"
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
