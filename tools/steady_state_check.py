"""Throughput over time: segments of 20 steps for 300 steps of bench.py's default step (is the 20-step > 100-step gap a clock ramp-down?)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench, scenes
sc = scenes.make_scene("surfel", 300000, 1920, 1080, seed=0, color_mode="precomp")
step, state = bench.make_step("surfel", sc, torch.device("cuda", 0))
for _ in range(10): step()
torch.cuda.synchronize()
seg = []
for s in range(15):
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    seg.append(round(20 / (time.perf_counter() - t0), 1))
print(json.dumps({"iters_per_s_per_20_step_segment": seg}))
