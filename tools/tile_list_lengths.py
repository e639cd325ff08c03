"""Distribution of tile-list lengths in the method pipelines (which path of the per-tile depth sort they take):
    python tools/tile_list_lengths.py [octree-pgsr|scaffold-2dgs]"""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from gsrast import rasterize as rz

which = sys.argv[1] if len(sys.argv) > 1 else "octree-pgsr"
seen = []
orig = rz.forward


def spy(variant, means3D, sh, colors_precomp, *a, **k):
    out = orig(variant, means3D, sh, colors_precomp, *a, **k)
    R, outs, radii, geom, binning, img = out
    settings = a[5] if len(a) > 5 else k["settings"]
    P = means3D.shape[0]
    T = ((settings.image_width + 15) // 16) * ((settings.image_height + 15) // 16)
    o = torch.empty((T, 2), dtype=torch.int32, device=means3D.device)
    rr = rz.debug_read(variant, 2, settings, P, 0, R, geom, binning, img, o).cpu().numpy().astype(np.int64)
    ln = np.maximum(rr[:, 1] - rr[:, 0], 0)
    seen.append({"variant": int(variant), "P": int(P), "R": int(R), "tiles": int(T), "mean": round(float(ln.mean()), 1), "median": float(np.median(ln)),
                 "p90": float(np.percentile(ln, 90)), "p99": float(np.percentile(ln, 99)), "max": int(ln.max()),
                 "tiles_le_256": round(float((ln <= 256).mean()), 4), "tiles_257_1024": round(float(((ln > 256) & (ln <= 1024)).mean()), 4),
                 "tiles_gt_1024": round(float((ln > 1024).mean()), 4),
                 "instances_in_tiles_gt_256": round(float(ln[ln > 256].sum() / max(1, ln.sum())), 4)})
    return out


rz.forward = spy
for m in list(sys.modules.values()):
    if m is not None and getattr(m, "__name__", "").startswith("diff_") and hasattr(m, "rz"):
        pass
if which == "octree-pgsr":
    import bench_pipeline_octree_pgsr as bp
    step, st = bp.build(types.SimpleNamespace(Na=74000, static=False), torch.device("cuda:0"))
else:
    import bench_pipeline as bp
    ns = types.SimpleNamespace(Na=72000, static=False, decode="hip", loss="bench", stop_after=None, graph=False, lod=False)
    step, st = bp.build(ns, torch.device("cuda:0"))
step()
torch.cuda.synchronize()
for s in seen:
    print(json.dumps(s))
