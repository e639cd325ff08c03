"""End-to-end scaffold-2dgs iteration on synthetic anchors (BASELINE.json configs[1] with its neural-Gaussian decode in front):
    prefilter (scaffold_filter.visible_filter) -> neural-Gaussian decode -> diff_surfel_rasterization fwd -> loss -> backward -> fused Adam.
--decode hip   : gsrast.decode (fused HIP, include/gsdecode.h)
--decode torch : the reference's torch op chain for the decode (tests/ref_decode_torch.py transcription), same rasterizer
Na anchors x k=10 offsets sized so that ~300k Gaussians reach the rasterizer at 1920x1080.  One JSON line."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hiprun                          # noqa: E402
import ref_decode_torch                # noqa: E402
import scenes                          # noqa: E402
import diff_surfel_rasterization as dsr   # noqa: E402
import scaffold_filter as sf           # noqa: E402
from gsrast import decode              # noqa: E402
from gsrast.losses import scaling_prod_mean, camera_ray_matrices, l1_plus_linear, l1_ssim, surfel_geo_loss  # noqa: E402
import ref_geo_torch                   # noqa: E402
import ref_loss_torch                  # noqa: E402
from gsrast.optim import Adam          # noqa: E402


def build(a, dev, seed=0):
    """-> (step, st): one scaffold-2dgs training iteration on a synthetic anchor scene; a has .decode, .loss, .Na and optionally .static
    (True: the sync-free, static-shape form of the same iteration -- decode with static_rows, so that it can be recorded into a HIP graph,
    gsrast.graphs.GraphedStep) and .lod
    (True: octree-2dgs, BASELINE configs[3]/[4] -- OctreeScene's level-of-detail mask + prefilter (gsr_octree_visible) in front of the same
    decode / surfel rasterizer / losses, anchors on 6 octree levels)."""
    W, H, k, A = 1920, 1080, 10, 32
    lod = bool(getattr(a, "lod", False))
    static = bool(getattr(a, "static", False))
    sc = scenes.make_scene("surfel", a.Na, W, H, seed=seed, color_mode="precomp")
    t = hiprun.to_dev(sc, dev)
    rs = hiprun.settings("surfel", t)
    fs = sf.GaussianRasterizationSettings(**rs._asdict())
    g = torch.Generator(device="cpu").manual_seed(7)
    anchor = t["means3D"].clone().requires_grad_(True)
    s2 = t["scales"]                                         # (Na,2) world-space sigma of the synthetic scene
    ext = s2.mean(dim=1, keepdim=True)
    scaling_log = torch.log(torch.cat([3.0 * ext.expand(-1, 3), 2.0 * s2, 2.0 * s2[:, :1]], dim=1)).requires_grad_(True)
    feat = torch.randn(a.Na, 32, generator=g).to(dev).requires_grad_(True)
    offset = (0.5 * torch.randn(a.Na, k, 3, generator=g)).to(dev).requires_grad_(True)
    rot_anchor = torch.nn.functional.normalize(torch.randn(a.Na, 4, generator=g), dim=1).to(dev)
    mlp = lambda i, o, act: torch.nn.Sequential(torch.nn.Linear(i, 32), torch.nn.ReLU(True), torch.nn.Linear(32, o), act).to(dev)
    torch.manual_seed(3)
    mlp_o, mlp_c, mlp_k = mlp(35, k, torch.nn.Tanh()), mlp(35, 7 * k, torch.nn.Identity()), mlp(35 + A, 3 * k, torch.nn.Sigmoid())
    emb = torch.nn.Embedding(4, A).to(dev)
    params = [anchor, scaling_log, feat, offset, emb.weight] + [p for m in (mlp_o, mlp_c, mlp_k) for p in m.parameters()]
    opt = (Adam(params, lr=1e-4, eps=1e-15) if os.environ.get("GSR_PIPE_TORCH_ADAM", "0") != "1" else torch.optim.Adam(params, lr=1e-4, eps=1e-15, fused=True))
    gt = torch.rand((3, H, W), generator=g).to(dev)
    N = float(W * H)
    gtn = torch.nn.functional.normalize(torch.randn((3, H, W), generator=g), dim=0)
    wmap = torch.zeros((11, H, W))
    wmap[0] = 0.01 / N; wmap[1] = 0.01 / N; wmap[2:5] = -0.05 * gtn / N; wmap[5] = 0.01 / N; wmap[6] = 100.0 / N
    wmap = wmap.to(dev)
    campos = t["campos"]
    wvt, fpt = t["viewmatrix"], t["projmatrix"]
    rm, nr = camera_ray_matrices(wvt, fpt, W, H)
    case = {"k": k, "dist_o": False, "dist_c": False, "dist_k": False}
    if lod:
        from gsrast import octree
        LEVELS, FORK = 6, 2.0
        level = torch.randint(0, LEVELS, (a.Na, 1), generator=g).to(dev)
        extra_level = torch.zeros(a.Na, device=dev)
        dist = (t["means3D"] - t["campos"]).norm(dim=1)
        standard_dist = float(dist.median()) * FORK ** 3.5     # the median anchor predicts level 3.5: levels 0..3 or 0..4 of 0..5 pass the mask
        voxel_size = float(ext.median()) * 8.0
    acc = {"opacity_accum": torch.zeros(a.Na, 1, device=dev), "anchor_demon": torch.zeros(a.Na, 1, device=dev),
           "offset_gradient_accum": torch.zeros(a.Na * k, 1, device=dev), "offset_denom": torch.zeros(a.Na * k, 1, device=dev)}
    st = {}

    stop = getattr(a, "stop_after", None)        # debugging aid (tools/debug_graph.py): return the intermediates of a prefix of the iteration

    def step():
        scaling = torch.exp(scaling_log)
        with torch.no_grad():                                 # prefilter_voxel (scaffold_scene.py:122-155)
            if lod:                                           # set_anchor_mask + prefilter_voxel of the Octree model in one call, no host sync
                vmask = octree.octree_visible(fs, anchor, level, scaling, rot_anchor, voxel_size, FORK, standard_dist, LEVELS, dist2level="round",
                                              extra_level=extra_level)["visible_mask"]
            else:
                radii = sf.GaussianRasterizer(fs).visible_filter(means3D=anchor, scales=scaling[:, :3], rotations=rot_anchor, cov3D_precomp=None)
                vmask = radii > 0
        app = emb.weight[1]
        if stop == "prefilter":
            return [vmask.clone(), decode.compact_visible(vmask, padded=True).clone()]
        if a.decode == "hip":
            vis_idx = decode.compact_visible(vmask, padded=True)   # once per iteration, shared by the decode and the statistics; no host sync
            count = None
            if static:
                xyz, color, opacity, scl, rot, nop, mask, count = decode.neural_gaussians(anchor, feat, offset, scaling, mlp_o, mlp_c, mlp_k, campos,
                                                                                         vis_idx=vis_idx, appearance=app, static_rows=True)
            else:
                # reference-shaped rows; the count is read as soon as the opacity head and its scan are done (deferred decode: the emit kernel is still
                # running while the host goes on to enqueue the rasterizer).  GSR_PIPE_DEFER=0: the synchronous call, for A/B
                out = decode.neural_gaussians(anchor, feat, offset, scaling, mlp_o, mlp_c, mlp_k, campos, vis_idx=vis_idx, appearance=app,
                                              deferred=os.environ.get("GSR_PIPE_DEFER", "1") != "0")
                xyz, color, opacity, scl, rot, nop, mask = out.finish() if isinstance(out, decode.PendingDecode) else out
        else:
            vis = torch.nonzero(vmask).view(-1)
            leaves = {"anchor": anchor, "feat": feat, "offset": offset, "scaling": scaling}
            par = {"W1o": mlp_o[0].weight, "b1o": mlp_o[0].bias, "W2o": mlp_o[2].weight, "b2o": mlp_o[2].bias,
                   "W1c": mlp_c[0].weight, "b1c": mlp_c[0].bias, "W2c": mlp_c[2].weight, "b2c": mlp_c[2].bias,
                   "W1k": mlp_k[0].weight, "b1k": mlp_k[0].bias, "W2k": mlp_k[2].weight, "b2k": mlp_k[2].bias, "app": app}
            o, _ = ref_decode_torch.decode_live(case, leaves, par, vis, campos)
            xyz, color, opacity, scl, rot = o["xyz"], o["color"], o["opacity"].view(-1, 1), o["scaling"], o["rot"]
        if stop == "decode":
            return [xyz.clone(), opacity.clone(), scl.clone(), rot.clone(), color.clone()] + ([count.clone()] if static else [])
        if static:                     # screen-space gradient carrier (only its .grad slot is used): one persistent zero leaf while shapes are static
            if st.get("m2") is None or st["m2"].shape != xyz.shape:
                st["m2"] = torch.zeros_like(xyz, requires_grad=True)
            means2D = st["m2"]; means2D.grad = None
        else:
            means2D = torch.zeros_like(xyz, requires_grad=True)
        img, rad, allmap = dsr.GaussianRasterizer(rs)(means3D=xyz, means2D=means2D, opacities=opacity, colors_precomp=color,
                                                      scales=scl[:, :2].contiguous(), rotations=rot)
        if stop == "raster":
            return [img.clone(), rad.clone(), allmap.clone()]
        # scaling_loss (scaffold_2dgs_scene.py:25: lambda_scaling * scaling.prod(dim=1).mean(), two columns for 2DGS): gsrast.losses.scaling_prod_mean --
        # value and gradient in one kernel; GSR_PIPE_TORCH_REG=1 keeps the torch chain (x*y on unbound columns: prod's backward synchronises the host
        # when an entry is 0, slices cost one zero-filled (P,3) gradient each)
        if os.environ.get("GSR_PIPE_TORCH_REG", "0") == "1" or a.decode != "hip":
            sx, sy, _sz = scl.unbind(dim=1)
            reg = 0.01 * ((sx * sy).sum() / count.to(torch.float32)[0] if (static and a.decode == "hip") else (sx * sy).mean())
        else:
            reg = scaling_prod_mean(scl, 0.01, cols=2, count=count if static else None, unit_upstream=True)
        if a.loss == "bench":
            loss = l1_plus_linear(img, gt, allmap, wmap) + reg
        elif a.loss == "full-hip":
            loss = l1_ssim(img, gt, 0.2, unit_upstream=True) + surfel_geo_loss(allmap, rm, nr, 0.0, 0.05, 100.0, unit_upstream=True)[0] + reg
        else:
            loss = ref_loss_torch.loss(img.unsqueeze(0), gt.unsqueeze(0), 0.2)[0] + ref_geo_torch.geo_loss(allmap, wvt, fpt, 0.0, 0.05, 100.0)[0] + reg
        if stop == "loss":
            return [loss.detach().clone()]
        loss.backward()
        if stop == "backward":
            out = [p.grad.clone() for p in params if p.grad is not None] + [means2D.grad.clone()]
            opt.zero_grad(set_to_none=True)
            return out
        if a.loss != "bench":                                    # densify(): training_statis every iteration (scaffold_gaussian.py:707-712)
            if a.decode == "hip":
                decode.training_stats_(acc["opacity_accum"], acc["anchor_demon"], acc["offset_gradient_accum"], acc["offset_denom"], means2D.grad,
                                       nop, rad > 0, mask, vis_idx=vis_idx)
            else:
                ref_decode_torch.training_statis(acc, k, means2D.grad, o["neural_opacity"].view(-1, 1), rad > 0, o["mask"], vmask)
        if stop == "stats":
            opt.zero_grad(set_to_none=True)
            return [acc["opacity_accum"].clone()]
        opt.step()
        if stop == "step":
            out = [p.detach().clone() for p in params]
            opt.zero_grad(set_to_none=True)
            return out
        opt.zero_grad(set_to_none=True)
        if "Nv" not in st:                       # once (first eager call): host reads for the report
            st["Nv"] = int(vmask.sum())
            st["P"] = int(count[0]) if (static and a.decode == "hip") else xyz.shape[0]
        st["rows"] = xyz.shape[0]
        return loss

    st["optimizers"] = [opt]

    return step, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--decode", default="hip", choices=["hip", "torch"])
    ap.add_argument("--loss", default="bench", choices=["bench", "full-hip", "full-torch"],
                    help="bench: L1 + linear aux (bench.py's loss); full-*: the reference's L1+SSIM + normal/dist regularisers + scaling loss, "
                         "fused HIP kernels or the reference's torch formulas")
    ap.add_argument("--Na", type=int, default=72000)
    ap.add_argument("--lod", action="store_true", help="octree-2dgs: level-of-detail mask + prefilter (use --Na 87000 for ~300k Gaussians)")
    ap.add_argument("--static", action="store_true", help="sync-free static-shape iteration (decode static_rows)")
    ap.add_argument("--graph", action="store_true", help="record the (static) iteration into a HIP graph and replay it (gsrast.graphs.GraphedStep)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    a = ap.parse_args()
    if a.graph:
        a.static = True
    step, st = build(a, torch.device("cuda:0"))
    if a.graph:
        from gsrast.graphs import GraphedStep
        step = GraphedStep(step, optimizers=st["optimizers"], warmup=max(3, a.warmup))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if a.graph:
        st["async_status"] = step.check()
    print(json.dumps({"pipeline": "octree-2dgs" if a.lod else "scaffold-2dgs", "mode": "graph" if a.graph else ("static" if a.static else "eager"),
                      "rows": st.get("rows"), "async_status": st.get("async_status"), "decode": a.decode, "loss": a.loss, "Na": a.Na, "Nv": st["Nv"], "P": st["P"], "steps": a.steps,
                      "ms_per_iter": 1e3 * dt / a.steps, "iters_per_s": a.steps / dt}))


if __name__ == "__main__":
    main()
