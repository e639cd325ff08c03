import cProfile, pstats, sys, os, types, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, os.path.join(ROOT,"tools"))
import bench_pipeline
step, st = bench_pipeline.build(types.SimpleNamespace(decode="hip", loss="full-hip", Na=72000), torch.device("cuda:0"))
for _ in range(10): step()
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
ps=pstats.Stats(pr); ps.sort_stats("tottime").print_stats(28)
