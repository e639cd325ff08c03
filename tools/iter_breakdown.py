"""GPU-kernel time vs wall time of one complete method iteration (scaffold-2dgs = configs[1], octree/vanilla-pgsr = configs[2]).
Runs the tools/bench_pipeline*.py iteration under torch.profiler (roctracer sees every HIP kernel of the process, including the
ones launched through the C ABI) and reports per iteration: wall ms, summed kernel ms, kernel launches, and the kernels by total time."""
import argparse
import collections
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def measure(step, dev, steps=20, warmup=8, top=14):
    from torch.profiler import profile, ProfilerActivity
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev); wall = (time.perf_counter() - t0) / steps
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA and not e.name.startswith(("Optimizer.", "ProfilerStep")):      # kernels / memsets only, no annotations
            t = getattr(e, "device_time_total", None)
            if t is None:
                t = e.cuda_time_total
            agg[e.name][0] += t; agg[e.name][1] += 1
    tot = sum(v[0] for v in agg.values()) / steps / 1e3
    n = sum(v[1] for v in agg.values()) / steps
    kernels = sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]
    return {"wall_ms": round(wall * 1e3, 4), "kernel_ms": round(tot, 4), "wall_over_kernel": round(wall * 1e3 / max(tot, 1e-9), 3),
            "launches_per_iter": round(n, 1),
            "top_kernels_ms": {k[:70]: round(v[0] / steps / 1e3, 4) for k, v in kernels},
            # every framework (at::native / rocclr) kernel by its full functor name: [launches per iteration, ms per iteration]
            "framework_kernels": {k[:260]: [round(v[1] / steps, 2), round(v[0] / steps / 1e3, 4)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])
                                  if ("at::native" in k or "rocclr" in k or k.startswith("Memset") or k.startswith("Memcpy"))}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="scaffold-2dgs", choices=["scaffold-2dgs", "octree-2dgs", "pgsr", "octree-pgsr"])
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.method in ("scaffold-2dgs", "octree-2dgs"):
        import bench_pipeline
        lod = a.method == "octree-2dgs"
        step, st = bench_pipeline.build(types.SimpleNamespace(decode="hip", loss="full-hip", Na=87000 if lod else 72000, lod=lod), dev)
    elif a.method == "octree-pgsr":
        import bench_pipeline_octree_pgsr
        step, st = bench_pipeline_octree_pgsr.build(types.SimpleNamespace(Na=74000), dev)
    else:
        import bench_pipeline_pgsr
        step, st = bench_pipeline_pgsr.build(types.SimpleNamespace(glue="hip", P=300000), dev)
    r = measure(step, dev, steps=a.steps)
    r["method"] = a.method
    print(json.dumps(r))


if __name__ == "__main__":
    main()
