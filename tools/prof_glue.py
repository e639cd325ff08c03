"""Per-iteration GPU time of the torch glue ops (zero fills, scalar multiplies, gradient accumulation adds ...) around the HIP kernels of a\ncomplete method iteration, grouped by op and input shape:  python tools/prof_glue.py octree|scaffold"""
import sys, os, types, torch
sys.path.insert(0, "tools")
which = sys.argv[1]
dev = torch.device("cuda:0")
if which == "octree":
    import bench_pipeline_octree_pgsr as b
    step, st = b.build(types.SimpleNamespace(Na=74000), dev)
else:
    import bench_pipeline as b
    step, st = b.build(types.SimpleNamespace(decode="hip", loss="full-hip", Na=72000), dev)
for _ in range(8): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    for _ in range(10): step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::zeros", "aten::zero_", "aten::fill_", "aten::zeros_like", "aten::add_", "aten::add", "aten::mul", "aten::copy_", "aten::sum", "aten::cat", "aten::empty", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::index", "aten::exp", "aten::mean", "aten::prod", "aten::gt", "aten::select_backward", "aten::slice_backward"):
        rows.append((e.device_time_total / 10.0, e.count / 10.0, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = 0
for t, n, k, s in rows[:40]:
    print(f"{t:8.1f} us/iter  x{n:4.1f}  {k:22s} {s}")
    tot += t
print("sum of listed", round(tot, 1), "us/iter")
