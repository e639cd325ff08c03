#!/usr/bin/env python3
"""Per-step host and device time of bench.py's timed loop right after its barrier (diagnostic for short --steps runs).

    python tools/step_series.py [--steps 40] [--warmup 5] [--prof 1]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench, gsrast, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--prof", type=int, default=1)
ap.add_argument("--stages-only", type=int, default=0)
ap.add_argument("--spin-kind", default="valu")
ap.add_argument("--spin-ms", type=float, default=0.0, help="busy the device this long before the warm-up")
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.make_scene("surfel", 300000, 1920, 1080, seed=0, color_mode="precomp")
step, state = bench.make_step("surfel", sc, dev)
if a.spin_ms > 0 and a.spin_kind == "hbm":
    x = torch.empty(1 << 28, device=dev); t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < a.spin_ms:
        x.mul_(1.0001); torch.cuda.synchronize()
elif a.spin_ms > 0:
    bench.clock_prewarm(dev, a.spin_ms)
for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
if a.stages_only:
    rows = []; tot = []
    for i in range(a.steps):
        gsrast.profile_enable(True)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record()
        torch.cuda.synchronize()
        p = gsrast.profile_read(); gsrast.profile_enable(False)
        rows.append({k: round(v[0] / max(1, v[1]), 4) for k, v in p.items()}); tot.append(round(e0.elapsed_time(e1), 3))
    for k in rows[0]:
        print(k, [r[k] for r in rows])
    print("step_total", tot)
    print("other", [round(t - sum(r.values()), 3) for t, r in zip(tot, rows)])
    sys.exit(0)
if a.prof:
    gsrast.profile_enable(True, stages=["blend_bwd"])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
host = []
t0 = time.perf_counter()
ev[0].record()
for i in range(a.steps):
    step()
    ev[i + 1].record()
    host.append(time.perf_counter())
torch.cuda.synchronize()
t1 = time.perf_counter()
d = [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(a.steps)]
h = [round(1e3 * (host[i] - (host[i - 1] if i else t0)), 3) for i in range(a.steps)]
print(json.dumps({"wall_ms_per_step": round(1e3 * (t1 - t0) / a.steps, 4), "device_ms": d, "host_ms": h,
                  "host_done_ms": round(1e3 * (host[-1] - t0), 3), "all_done_ms": round(1e3 * (t1 - t0), 3)}))
# per-step stage times (each step synchronised: kernel times only, no pipelining)
rows = []
for i in range(a.steps):
    gsrast.profile_enable(True)
    step()
    torch.cuda.synchronize()
    p = gsrast.profile_read()
    gsrast.profile_enable(False)
    rows.append({k: round(v[0] / max(1, v[1]), 4) for k, v in p.items()})
for k in rows[0]:
    print(k, [r[k] for r in rows])
