"""Host-side (Python) profile of one scaffold-2dgs iteration: cProfile over 100 steps, functions by own time and the gsrast wrappers by
cumulative time.  The autograd backward runs on torch's engine thread, so its Python frames are not attributed here (see run_backward)."""
import cProfile, pstats, sys, os, types, io, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_pipeline
step, st = bench_pipeline.build(types.SimpleNamespace(decode="hip", loss="full-hip", Na=72000), torch.device("cuda:0"))
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
for key, n in (("tottime", 22), ("cumtime", 30)):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n); print(s.getvalue()[:6000])
