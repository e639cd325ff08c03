"""End-to-end octree-pgsr iteration after step 7000 on synthetic anchors (BASELINE.json configs[2]: OctreePGSRScene.get_train_loss_dict,
gssr/scene/octree_pgsr_scene.py:26-45), for the view camera AND the neighbour camera:
    level-of-detail mask + frustum prefilter (gsr_octree_visible) -> neural-Gaussian decode -> per-Gaussian all_map ->
    diff_plane_rasterization fwd
then L1+SSIM + single-view normal loss + multi-view geometric / NCC losses + scaling loss -> backward (both renders, both decodes) ->
training statistics -> fused Adam.  Na anchors x k=10 offsets on 6 octree levels, sized so that ~300k Gaussians reach the rasterizer
per camera at 1920x1080.  One JSON line."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hiprun                          # noqa: E402
import mv_cases                        # noqa: E402
import scenes                          # noqa: E402
import diff_plane_rasterization as dpr   # noqa: E402
import scaffold_filter as sf           # noqa: E402
from bench_pipeline_pgsr import cam_of   # noqa: E402
from gsrast import decode, octree      # noqa: E402
from gsrast.losses import scaling_prod_mean, l1_ssim, multiview_cfg, plane_geo_loss, plane_multiview_loss, plane_losses  # noqa: E402
from gsrast.plane_prep import plane_input_all_map  # noqa: E402
from gsrast.optim import Adam, shadow_parameters          # noqa: E402


def build(a, dev, seed=0):
    """-> (step, st): one octree-pgsr training iteration (step > 7000); a has .Na and optionally .static (the sync-free static-shape form:
    decode with static_rows, recordable into a HIP graph -- gsrast.graphs.GraphedStep)."""
    static = bool(getattr(a, "static", False))
    defer = os.environ.get("GSR_PIPE_DEFER", "1") != "0"       # A/B: "0" = each render reads its decode's count before anything else is enqueued
    W, H, k, A, LEVELS, FORK = 1920, 1080, 10, 32, 6, 2.0
    sc = scenes.make_scene("plane", a.Na, W, H, seed=seed, color_mode="precomp")
    t = hiprun.to_dev(sc, dev)
    cam2 = scenes.make_camera(W, H, W / (2 * sc["tanfovx"]), H / (2 * sc["tanfovy"]), yaw_deg=3.0, t=(-0.15, 0.02, 0.0))
    t2 = dict(t); t2.update({n: torch.tensor(cam2[n], device=dev) for n in ("viewmatrix", "projmatrix", "campos")})
    views = []
    for tt in (t, t2):
        views.append((tt, hiprun.settings("plane", tt), sf.GaussianRasterizationSettings(**hiprun.settings("ewa", tt)._asdict())))
    g = torch.Generator(device="cpu").manual_seed(7)
    anchor = t["means3D"].clone().requires_grad_(True)
    s3 = t["scales"]                                          # (Na,3) world-space sigma of the synthetic scene (one axis flat)
    ext = s3.max(dim=1, keepdim=True)[0]
    scaling_log = torch.log(torch.cat([3.0 * ext.expand(-1, 3), 2.0 * s3], dim=1)).requires_grad_(True)
    feat = torch.randn(a.Na, 32, generator=g).to(dev).requires_grad_(True)
    offset = (0.5 * torch.randn(a.Na, k, 3, generator=g)).to(dev).requires_grad_(True)
    rot_anchor = torch.nn.functional.normalize(torch.randn(a.Na, 4, generator=g), dim=1).to(dev)
    level = torch.randint(0, LEVELS, (a.Na, 1), generator=g).to(dev).to(torch.int32)      # int32 as the kernels read it (an int64 buffer costs a conversion launch per render)
    extra_level = torch.zeros(a.Na, device=dev)
    dist = (t["means3D"] - t["campos"]).norm(dim=1)
    standard_dist = float(dist.median()) * FORK ** 3.5         # median anchor predicts level 3.5: levels 0..3 or 0..4 of 0..5 pass the mask
    voxel_size = float(ext.median()) * 8.0
    mlp = lambda i, o, act: torch.nn.Sequential(torch.nn.Linear(i, 32), torch.nn.ReLU(True), torch.nn.Linear(32, o), act).to(dev)
    torch.manual_seed(3)
    mlp_o, mlp_c, mlp_k = mlp(35, k, torch.nn.Tanh()), mlp(35, 7 * k, torch.nn.Identity()), mlp(35 + A, 3 * k, torch.nn.Sigmoid())
    emb = torch.nn.Embedding(4, A).to(dev)
    params = [anchor, scaling_log, feat, offset, emb.weight] + [p for m in (mlp_o, mlp_c, mlp_k) for p in m.parameters()]
    opt = (Adam(params, lr=1e-4, eps=1e-15) if os.environ.get("GSR_PIPE_TORCH_ADAM", "0") != "1" else torch.optim.Adam(params, lr=1e-4, eps=1e-15, fused=True))
    # The neighbour camera's pass reads a second set of leaves over the same storage, and the optimizer adds the two passes' gradients inside its
    # update kernel: without it autograd sums them with one `add` launch per parameter tensor (27 per iteration, 130 us).  GSR_PIPE_SHADOWS=0: one set.
    first = {"anchor": anchor, "scaling_log": scaling_log, "feat": feat, "offset": offset, "emb": emb, "mlp_o": mlp_o, "mlp_c": mlp_c, "mlp_k": mlp_k}
    second = first
    if os.environ.get("GSR_PIPE_SHADOWS", "1") != "0" and isinstance(opt, Adam):
        second = shadow_parameters(first)
        opt.add_shadows(first, second)
    gt = torch.rand((3, H, W), generator=g).to(dev)
    gray1 = gt.mean(0, keepdim=True).contiguous(); gray2 = torch.rand((1, H, W), generator=g).to(dev)
    c1, c2 = cam_of(t, W, H), cam_of(t2, W, H)
    K1 = torch.tensor([[c1["Fx"], 0, c1["Cx"]], [0, c1["Fy"], c1["Cy"]], [0, 0, 1]], device=dev)
    rm1 = torch.inverse(K1.double().t()).float()
    weight = torch.rand((H, W), generator=g).to(dev)
    mcfg = multiview_cfg(mv_cases.cam_ns(c1), mv_cases.cam_ns(c2), W, H, near_size=(W, H))
    acc = {"opacity_accum": torch.zeros(a.Na, 1, device=dev), "anchor_demon": torch.zeros(a.Na, 1, device=dev),
           "offset_gradient_accum": torch.zeros(a.Na * k, 1, device=dev), "offset_denom": torch.zeros(a.Na * k, 1, device=dev)}
    st = {}
    carriers = {}

    def render_begin(view, cam_id, L):
        """LOD mask + prefilter + decode, enqueued; in the reference-shaped (eager) mode the decode is DEFERRED: its count is read in render_finish,
        after the other camera's decode has been enqueued behind it (gsrast.decode.PendingDecode)."""
        tt, rs, fs = view
        anchor, feat, offset, emb, mlp_o, mlp_c, mlp_k = L["anchor"], L["feat"], L["offset"], L["emb"], L["mlp_o"], L["mlp_c"], L["mlp_k"]
        scaling = torch.exp(L["scaling_log"])
        vis = octree.octree_visible(fs, anchor, level, scaling, rot_anchor, voxel_size, FORK, standard_dist, LEVELS, dist2level="round",
                                    extra_level=extra_level)   # set_anchor_mask + prefilter_voxel, no host sync
        vis_idx = decode.compact_visible(vis["visible_mask"], padded=True)
        out = decode.neural_gaussians(anchor, feat, offset, scaling, mlp_o, mlp_c, mlp_k, tt["campos"], vis_idx=vis_idx, appearance=emb.weight[cam_id],
                                      static_rows=static, deferred=not static and defer)
        return view, cam_id, vis, vis_idx, out

    def render(view, cam_id, L):
        return render_finish(render_begin(view, cam_id, L))

    def render_finish(begun):
        view, cam_id, vis, vis_idx, out = begun
        tt, rs, fs = view
        if isinstance(out, decode.PendingDecode):
            out = out.finish()
        xyz, color, opacity, scl, rot, nop, mask = out[:7]
        count = out[7] if static else None
        am = plane_input_all_map(xyz, rot, scl, tt["viewmatrix"], tt["campos"])
        # screen-space gradient carriers: the rasterizer only uses their .grad slot.  Static shapes: two persistent zero leaves per camera (no fill per
        # iteration); dynamic shapes: fresh ones, as the reference makes them (pgsr_scene.py: torch.zeros_like(means3D, requires_grad=True) + 0).
        # Screen-space gradient carriers: the rasterizer only uses their .grad slot and never reads or writes their values, so two persistent zero
        # buffers per camera serve every iteration -- static shapes: the leaves themselves; reference shapes (P changes every iteration): fresh leaves
        # that VIEW the first P rows of a zero buffer grown on demand (the reference fills two (P,3) tensors per render, pgsr_scene.py:287-288).
        if static:
            key = (cam_id, xyz.shape[0])
            if key not in carriers:
                carriers[key] = (torch.zeros_like(xyz, requires_grad=True), torch.zeros_like(xyz, requires_grad=True))
            m2, m2a = carriers[key]
            m2.grad = None; m2a.grad = None
        else:
            buf = carriers.get(cam_id)
            if buf is None or buf[0].shape[0] < xyz.shape[0]:
                n = int(xyz.shape[0] * 1.25) + 1024
                buf = carriers[cam_id] = (torch.zeros(n, 3, device=xyz.device), torch.zeros(n, 3, device=xyz.device))
            m2 = buf[0][: xyz.shape[0]].detach().requires_grad_(True); m2a = buf[1][: xyz.shape[0]].detach().requires_grad_(True)
        img, radii, obs, oam, pd = dpr.GaussianRasterizer(rs)(means3D=xyz, means2D=m2, means2D_abs=m2a, opacities=opacity, colors_precomp=color,
                                                             scales=scl, rotations=rot, all_map=am)
        return img, radii, oam, pd, scl, m2, nop, mask, vis_idx, vis["visible_mask"], count

    def step():
        b1, b2 = render_begin(views[0], 1, first), render_begin(views[1], 2, second)       # both decodes in flight before the first count is read
        img, radii, oam, pd, scl, m2, nop, mask, vis_idx, vmask, count = render_finish(b1)
        pd2 = render_finish(b2)[3]
        if os.environ.get("GSR_PIPE_TORCH_REG", "0") == "1":
            sx, sy, sz = scl.unbind(dim=1)                # x*y*z on unbound columns (backward = ONE stack), not prod(dim=1): prod's backward
            vol = sx * sy * sz                            # synchronises (nonzero) when an entry is 0, and not scl[:, i]: one zero-filled (P,3) per slice
            reg = 0.01 * (vol.sum() / count.to(torch.float32)[0] if static else vol.mean())
        else:                                             # scaling_loss (octree_pgsr_scene.py:23), value and gradient in one kernel
            reg = scaling_prod_mean(scl, 0.01, count=count if static else None, unit_upstream=True)
        nrm, geo, ncc = plane_losses(pd, pd2, oam, gray1, gray2, mcfg, rm1, weight, 0.015, 0.03, 0.15)      # one node: gradients to pd / oam leave it summed
        # the total loss is the SUM of five terms; its backward sends 1 to each: five roots with unit gradients are the same backward pass without the
        # four scalar add launches (and their autograd nodes) of `(a + b + c + d + e).backward()`
        roots = [l1_ssim(img, gt, 0.2, unit_upstream=True), nrm, reg, geo, ncc]
        if "ones" not in st:
            st["ones"] = [torch.ones_like(r) for r in roots]
        torch.autograd.backward(roots, st["ones"])
        decode.training_stats_(acc["opacity_accum"], acc["anchor_demon"], acc["offset_gradient_accum"], acc["offset_denom"], m2.grad, nop, radii > 0,
                               mask, vis_idx=vis_idx)
        opt.step(); opt.zero_grad(set_to_none=True)
        if "P" not in st:
            st["P"] = int(mask.sum()); st["Nv"] = int(vmask.sum())

    st["optimizers"] = [opt]
    return step, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--static", action="store_true", help="sync-free static-shape iteration (decode static_rows)")
    ap.add_argument("--graph", action="store_true", help="record the (static) iteration into a HIP graph and replay it")
    ap.add_argument("--Na", type=int, default=74000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
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
    print(json.dumps({"pipeline": "octree-pgsr (step > 7000: LOD mask + prefilter + decode + plane render, twice; single-view + multi-view losses)",
                      "mode": "graph" if a.graph else ("static" if a.static else "eager"), "Na": a.Na, "Nv": st["Nv"], "P": st["P"], "steps": a.steps, "ms_per_iter": round(1e3 * dt / a.steps, 3),
                      "iters_per_s": round(a.steps / dt, 1)}))


if __name__ == "__main__":
    main()
