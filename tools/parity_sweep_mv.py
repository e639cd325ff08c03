"""Randomised HIP-vs-oracle sweep of the PGSR multi-view losses, the plane all_map prep and the anchor statistics (many seeds / shapes);
prints one JSON line with the worst deviations.  Complements tools/parity_sweep.py (rasterizers)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mv_cases, oracle_multiview as om, oracle_decode
from test_gpu_mvloss import _run_hip, _rel
from gsrast.plane_prep import plane_input_all_map
from gsrast.decode import training_stats_
DEV = "cuda:0"
t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)
worst = {"mask_flips": 0, "noise": 0.0, "geo_loss_rel": 0.0, "g_depth": 0.0, "g_near": 0.0, "ncc_mask_flips": 0, "ncc_q99": 0.0, "ncc_loss_rel": 0.0,
         "g_normal": 0.0, "g_dist": 0.0, "allmap": 0.0, "allmap_dx": 0.0, "allmap_dq": 0.0, "stats_mismatch": 0}
r = np.random.default_rng(123)
n_mv = 0
for seed in range(24):
    W = int(r.integers(40, 260)); H = int(r.integers(30, 200))
    case = mv_cases.plane_pair(W=W, H=H, seed=100 + seed, amp=float(r.uniform(0.0, 0.3)), near_yaw=float(r.uniform(-15, 15)),
                               near_t=(float(r.uniform(-0.5, 0.5)), float(r.uniform(-0.1, 0.1)), float(r.uniform(-0.1, 0.1))), tex=float(r.uniform(1, 6)))
    patch = int(r.integers(1, 5))
    cfg = om.make_cfg(W, H, case["view"], case["near"], patch=patch)
    og = om.geo(cfg, case["plane_depth"], case["near_plane_depth"])
    geo, ncc, aux, g = _run_hip(case, patch=patch)
    fl = aux["d_mask"].reshape(-1) != og["dmask"].astype(bool)
    worst["mask_flips"] = max(worst["mask_flips"], int(fl.sum()))
    both = ~fl & og["dmask"].astype(bool)
    if both.any():
        worst["noise"] = max(worst["noise"], float(np.abs(aux["pixel_noise"].reshape(-1) - og["noise"])[both].max()))
    cnt = og["stats"][1]
    if cnt > 0 and not fl.any():
        worst["geo_loss_rel"] = max(worst["geo_loss_rel"], abs(geo / (0.03 * og["stats"][0] / cnt) - 1))
        worst["g_depth"] = max(worst["g_depth"], _rel(g["plane_depth"].reshape(-1) / 2, 0.03 / cnt * og["g_depth"]))
        worst["g_near"] = max(worst["g_near"], _rel(g["near_plane_depth"].reshape(-1) / 2, 0.03 / cnt * og["g_near"]))
    idx = np.nonzero(og["dmask"])[0].astype(np.int32)
    if idx.size:
        on = om.ncc(cfg, idx, og["weight"], case["rendered_normal"], case["rendered_distance"], case["gray"], case["near_gray"])
        _, ncc2, aux2, g2 = _run_hip(case, indices=idx, patch=patch)
        mf = aux2["ncc_mask"] != on["mask"].astype(bool)
        worst["ncc_mask_flips"] = max(worst["ncc_mask_flips"], int(mf.sum()))
        e = np.abs(aux2["ncc"] - on["ncc"])[~mf]
        worst["ncc_q99"] = max(worst["ncc_q99"], float(np.quantile(e, 0.99)))
        if on["stats"][1] > 0 and not mf.any():
            sc = 0.15 / on["stats"][1]
            worst["ncc_loss_rel"] = max(worst["ncc_loss_rel"], abs(ncc2 / (0.15 * on["stats"][0] / on["stats"][1]) - 1))
            worst["g_normal"] = max(worst["g_normal"], _rel(g2["rendered_normal"].reshape(3, -1) / 3, sc * on["g_normal"]))
            worst["g_dist"] = max(worst["g_dist"], _rel(g2["rendered_distance"].reshape(-1) / 3, sc * on["g_dist"]))
    n_mv += 1
cam = mv_cases.plane_pair(W=64, H=48, seed=0)
import scenes
V = scenes.make_camera(640, 480, 500.0, 500.0, yaw_deg=17.0, t=(0.3, -0.2, 0.4))
for seed in range(12):
    P = int(r.integers(1, 50000))
    xyz = r.uniform(-6, 6, (P, 3)).astype(np.float32); q = r.normal(0, 1, (P, 4)).astype(np.float32) * r.uniform(0.2, 3.0)
    sc = np.exp(r.normal(-2, 1, (P, 3))).astype(np.float32); dL = r.normal(0, 1, (P, 5)).astype(np.float32)
    oam, odx, odq = om.plane_allmap(xyz, q, sc, V["viewmatrix"], V["campos"], dL)
    x = t(xyz).requires_grad_(True); qq = t(q).requires_grad_(True)
    am = plane_input_all_map(x, qq, t(sc), t(V["viewmatrix"]), t(V["campos"]))
    (am * t(dL)).sum().backward()
    worst["allmap"] = max(worst["allmap"], float(np.abs(am.detach().cpu().numpy() - oam).max() / np.abs(oam).max()))
    worst["allmap_dx"] = max(worst["allmap_dx"], float(np.abs(x.grad.cpu().numpy() - odx).max() / np.abs(odx).max()))
    worst["allmap_dq"] = max(worst["allmap_dq"], float(np.abs(qq.grad.cpu().numpy() - odq).max() / np.abs(odq).max()))
for seed in range(12):
    Na = int(r.integers(1, 30000)); k = int(r.integers(1, 17))
    vis = r.uniform(size=Na) < r.uniform(0.05, 1.0); Nv = int(vis.sum())
    if Nv == 0:
        continue
    nop = np.tanh(r.normal(0, 1, Nv * k)).astype(np.float32); sel = nop > 0; P = int(sel.sum())
    upd = r.uniform(size=P) < 0.7; grad = r.normal(0, 1, (max(P, 1), 3)).astype(np.float32)[:P]
    host = [r.uniform(0, 1, Na).astype(np.float32), r.integers(0, 9, Na).astype(np.float32), r.uniform(0, 1, Na * k).astype(np.float32),
            r.integers(0, 9, Na * k).astype(np.float32)]
    devt = [t(a.reshape(-1, 1)) for a in host]
    if P == 0:
        continue
    oracle_decode.training_stats(np.nonzero(vis)[0], k, nop, sel, upd, grad, *host)
    training_stats_(*devt, t(grad), t(nop), t(upd), t(sel), anchor_visible_mask=t(vis))
    worst["stats_mismatch"] += int(sum(not np.array_equal(a, b.cpu().numpy().reshape(-1)) for a, b in zip(host, devt)))
print(json.dumps({"multiview_cases": n_mv, "worst": {k: (float(v) if not isinstance(v, int) else v) for k, v in worst.items()}}))
