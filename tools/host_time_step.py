"""Host time of bench.py's headline step: the same step on a scene so small that the device is never the bottleneck (P = 2000, 160x112), so wall time per step = what the host
needs to enqueue one iteration -- the figure that decides whether a box's host keeps up with the 0.73 ms of kernels of the 300k / 1080p step (the eager headline follows the host:
1279-1350 it/s on three pool boxes in round 6).  cProfile of the main thread on top.   python tools/host_time_step.py [variant]"""
import cProfile, io, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd"))
import torch
import bench
from gsrast import workloads as scenes
variant = sys.argv[1] if len(sys.argv) > 1 else "surfel"
sc = scenes.make_scene(variant, 2000, 160, 112, seed=0, color_mode="precomp")
step, state = bench.make_step(variant, sc, torch.device("cuda", 0))
for _ in range(50): step()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 200)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:7000])
print(json.dumps({"variant": variant, "host_ms_per_step": round(1e3 * best, 4), "cpu": os.cpu_count()}))
