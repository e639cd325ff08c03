"""Host time per octree-pgsr iteration with the device queue never full: python tools/host_time_octree.py (GSR_PIPE_SHADOWS=0/1)."""
import os, sys, time, types, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_pipeline_octree_pgsr as bp
a = types.SimpleNamespace(Na=74000, static=False)
step, st = bp.build(a, torch.device("cuda:0"))
for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(30):
    step(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print("ms per synchronised iteration", round(1e3 * dt / 30, 3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
