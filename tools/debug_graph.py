"""Bisects the graph-replayed scaffold-2dgs iteration: records growing prefixes of it and compares each replay with the eager result."""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_pipeline
dev = torch.device("cuda:0")
for stop in ("prefilter", "decode", "raster", "loss", "backward"):
    a = types.SimpleNamespace(decode="hip", loss="full-hip", Na=9000, static=True, stop_after=stop)
    step, st = bench_pipeline.build(a, dev)
    ref = step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    for rep in range(3):
        g.replay(); torch.cuda.synchronize()
        res = []
        for o, r in zip(out, ref):
            o = o.float(); r = r.float()
            bad = ~torch.isfinite(o)
            res.append("%.3e%s" % (float((o - r).abs().max()) if o.numel() else 0.0, " NONFINITE" if bool(bad.any()) else ""))
        print(stop, "replay", rep, "max abs diff per output:", res, flush=True)

print("---- suffixes with the optimizer inside the graph")
from gsrast import rasterize as rz
for stop in ("stats", "step", None):
    outs = {}
    for mode in ("eager", "graph"):
        a = types.SimpleNamespace(decode="hip", loss="full-hip", Na=9000, static=True, stop_after=stop)
        step, st = bench_pipeline.build(a, dev)
        opt = st["optimizers"][0]
        pars = [p for g in opt.param_groups for p in g["params"]]
        if mode == "eager":
            for _ in range(3 + 3): step()
        else:
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3): step()
            torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
            rz.async_status_reset()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            for rep in range(3):
                opt.prepare_replay(); g.replay(); torch.cuda.synchronize()
                print(stop, "replay", rep, "status", rz.async_status(), "finite", all(bool(torch.isfinite(p).all()) for p in pars), flush=True)
        torch.cuda.synchronize()
        outs[mode] = [p.detach().clone() for p in pars]
    print(stop, "param diff eager vs graph:", ["%.2e" % float((a_ - b_).abs().max()) for a_, b_ in zip(outs["eager"], outs["graph"])], flush=True)
