"""Is bench.py's step loop GPU-bound?  Compares the host time needed to ENQUEUE K steps with the wall time to finish them."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench, scenes, gsrast
out = {}
for variant in ("surfel", "ewa", "plane"):
    sc = scenes.make_scene(variant, 300000, 1920, 1080, seed=0, color_mode="precomp")
    step, state = bench.make_step(variant, sc, torch.device("cuda", 0))
    for _ in range(10): step()
    torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out[variant] = {"enqueue_ms_per_step": round(1e3 * (t1 - t0) / K, 3), "total_ms_per_step": round(1e3 * (t2 - t0) / K, 3)}
print(json.dumps(out))
