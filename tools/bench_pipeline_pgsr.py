"""End-to-end PGSR iteration after step 7000 on a synthetic scene (two cameras, P Gaussians, 1920x1080):
    activations -> per-Gaussian all_map -> diff_plane_rasterization fwd (view) -> same for the neighbour camera ->
    L1+SSIM + single-view normal loss + multi-view geometric / NCC losses -> backward -> fused Adam.
--glue hip   : gsrast.plane_prep / gsrast.losses (fused HIP kernels) around the HIP rasterizer
--glue torch : the reference's torch op chains (tests/ref_*_torch.py restatements, each checked against reference-run fixtures) around
               the SAME HIP rasterizer -- what a user gets by swapping only the rasterizer extension.
One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hiprun                          # noqa: E402
import mv_cases                        # noqa: E402
import ref_geo_torch                   # noqa: E402
import ref_loss_torch                  # noqa: E402
import ref_mv_torch                    # noqa: E402
import scenes                          # noqa: E402
import diff_plane_rasterization as dpr   # noqa: E402
from gsrast.losses import l1_ssim, multiview_cfg, plane_geo_loss, plane_multiview_loss, plane_losses  # noqa: E402
from gsrast.plane_prep import plane_input_all_map  # noqa: E402
from gsrast.activations import gaussian_activations  # noqa: E402
from gsrast.optim import shadow_parameters  # noqa: E402
from gsrast.optim import Adam          # noqa: E402


def q2m(q):
    r, i, j, k = torch.unbind(q, -1); two_s = 2.0 / (q * q).sum(-1)
    return torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r), two_s * (i * j + k * r),
                        1 - two_s * (i * i + k * k), two_s * (j * k - i * r), two_s * (i * k - j * r), two_s * (j * k + i * r),
                        1 - two_s * (i * i + j * j)), -1).reshape(-1, 3, 3)


def torch_all_map(xyz, rot, scl, V, cpos):
    R = q2m(rot)
    idx = scl.min(dim=-1)[1][..., None, None].expand(-1, 3, -1)
    n = R.gather(2, idx).squeeze(2)
    neg = (n * (cpos - xyz)).sum(-1) < 0.0
    n = torch.where(neg[:, None], -n, n)
    ln = n @ V[:3, :3]
    pc = xyz @ V[:3, :3] + V[3, :3]
    am = torch.zeros(xyz.shape[0], 5, device=xyz.device)
    am[:, :3] = ln; am[:, 3] = 1.0; am[:, 4] = (ln * pc).sum(-1).abs()
    return am


def cam_of(t, W, H):
    V = t["viewmatrix"].double().cpu().numpy()
    return dict(R=V[:3, :3].copy(), T=V[3, :3].copy(), Fx=W / (2 * float(t["tanfovx"])), Fy=H / (2 * float(t["tanfovy"])), Cx=W / 2.0, Cy=H / 2.0)


def build(a, dev):
    """-> (step, st): one PGSR training iteration after step 7000 (two plane renders + single-view + multi-view losses); a has .glue, .P."""
    W, H = 1920, 1080
    sc = scenes.make_scene("plane", a.P, W, H, seed=0, color_mode="precomp")
    t = hiprun.to_dev(sc, dev)
    fx = W / (2 * sc["tanfovx"])
    cam2 = scenes.make_camera(W, H, fx, H / (2 * sc["tanfovy"]), yaw_deg=3.0, t=(-0.15, 0.02, 0.0))
    t2 = dict(t); t2.update({k: torch.tensor(cam2[k], device=dev) for k in ("viewmatrix", "projmatrix", "campos")})
    rs1, rs2 = hiprun.settings("plane", t), hiprun.settings("plane", t2)
    g = torch.Generator(device="cpu").manual_seed(7)
    xyz = t["means3D"].clone().requires_grad_(True)
    scl_log = torch.log(t["scales"]).requires_grad_(True)
    rot_raw = t["rotations"].clone().requires_grad_(True)
    op_raw = torch.logit(t["opacities"].clamp(1e-4, 1 - 1e-4)).requires_grad_(True)
    col = t["colors_precomp"].clone().requires_grad_(True)
    opt = (Adam([xyz, scl_log, rot_raw, op_raw, col], lr=1e-4, eps=1e-15) if os.environ.get("GSR_PIPE_TORCH_ADAM", "0") != "1" else torch.optim.Adam([xyz, scl_log, rot_raw, op_raw, col], lr=1e-4, eps=1e-15, fused=True))
    # The neighbour camera's pass reads a second set of leaves over the same storage (and runs its own activation kernel); the optimizer adds the two passes'
    # gradients inside its update kernel.  With ONE set autograd sums the two renders' contributions with one `add` launch per tensor the passes share
    # (xyz, scaling, rotation, opacity, colour: five of the nine adds of round 4's iteration).  GSR_PIPE_SHADOWS=0: one set.
    first = [xyz, scl_log, rot_raw, op_raw, col]
    second = first
    if a.glue == "hip" and os.environ.get("GSR_PIPE_SHADOWS", "1") != "0" and isinstance(opt, Adam):
        second = shadow_parameters(first)
        opt.add_shadows(first, second)
    gt = torch.rand((3, H, W), generator=g).to(dev)
    gray1 = gt.mean(0, keepdim=True).contiguous(); gray2 = torch.rand((1, H, W), generator=g).to(dev)
    c1, c2 = cam_of(t, W, H), cam_of(t2, W, H)
    K1 = torch.tensor([[c1["Fx"], 0, c1["Cx"]], [0, c1["Fy"], c1["Cy"]], [0, 0, 1]], device=dev)
    rm1 = torch.inverse(K1.double().t()).float()
    weight = torch.rand((H, W), generator=g).to(dev)                      # the detached image-gradient weight map (cached per camera)
    mcfg = multiview_cfg(mv_cases.cam_ns(c1), mv_cases.cam_ns(c2), W, H, near_size=(W, H))
    st = {"P": a.P}
    carriers = {}

    def render(rs, tt, means, scl, rot, op, col):
        V, cpos = tt["viewmatrix"], tt["campos"]
        am = plane_input_all_map(means, rot, scl, V, cpos) if a.glue == "hip" else torch_all_map(means, rot, scl, V, cpos)
        if a.glue == "hip":      # the rasterizer only uses the carriers' .grad slot: two persistent zero leaves per camera instead of two fills per render
            key = id(rs)
            if key not in carriers:
                carriers[key] = (torch.zeros_like(means, requires_grad=True), torch.zeros_like(means, requires_grad=True))
            m2, m2a = carriers[key]; m2.grad = None; m2a.grad = None
        else:
            m2 = torch.zeros_like(means, requires_grad=True); m2a = torch.zeros_like(means, requires_grad=True)
        return dpr.GaussianRasterizer(rs)(means3D=means, means2D=m2, means2D_abs=m2a, opacities=op, colors_precomp=col, scales=scl, rotations=rot,
                                          all_map=am)

    def step():
        if a.glue == "hip":              # get_scaling / get_rotation / get_opacity (vanilla_gaussian.py:250-269) as one kernel each way
            scl, rot, op = gaussian_activations(scl_log, rot_raw, op_raw)
        else:
            scl = torch.exp(scl_log); rot = torch.nn.functional.normalize(rot_raw); op = torch.sigmoid(op_raw)
        img, radii, obs, oam, pd = render(rs1, t, xyz, scl, rot, op, col)
        if second is first:
            _, _, _, _, pd2 = render(rs2, t2, xyz, scl, rot, op, col)
        else:
            scl2, rot2, op2 = gaussian_activations(second[1], second[2], second[3])
            _, _, _, _, pd2 = render(rs2, t2, second[0], scl2, rot2, op2, second[4])
        if a.glue == "hip":
            nrm, geo, ncc = plane_losses(pd, pd2, oam, gray1, gray2, mcfg, rm1, weight, 0.015, 0.03, 0.15)      # one node: gradients to pd / oam leave it summed
            # sum of four terms -> four roots with unit gradients: the same backward pass without the scalar add launches
            roots = [l1_ssim(img, gt, 0.2, unit_upstream=True), nrm, geo, ncc]
            if "ones" not in st:
                st["ones"] = [torch.ones_like(r) for r in roots]
            torch.autograd.backward(roots, st["ones"])
            opt.step(); opt.zero_grad(set_to_none=True)
            return
        else:
            loss = ref_loss_torch.loss(img.unsqueeze(0), gt.unsqueeze(0), 0.2)[0] + ref_geo_torch.plane_geo_loss(pd.squeeze(0), oam, K1, weight, 0.015)[0]
            # the reference draws its <= 102400 samples with np.random.choice on the host; here a device-side draw so that only the op chain is timed
            geo, ncc = ref_mv_torch.multiview_loss(pd, pd2, oam[0:3], oam[4:5], gray1, gray2, c1, c2, indices=st.get("idx"))
        (loss + geo + ncc).backward()
        opt.step(); opt.zero_grad(set_to_none=True)

    st["optimizers"] = [opt]
    if a.glue == "torch":                                                 # a fixed sample set of the reference's size
        with torch.no_grad():
            st["idx"] = torch.randperm(W * H, device=dev)[:102400]
    return step, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--glue", default="hip", choices=["hip", "torch"])
    ap.add_argument("--P", type=int, default=300000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    step, st = build(a, torch.device("cuda:0"))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"pipeline": "pgsr (step > 7000: two renders + single-view + multi-view losses)", "glue": a.glue, "P": a.P, "steps": a.steps,
                      "ms_per_iter": round(1e3 * dt / a.steps, 3), "iters_per_s": round(a.steps / dt, 1)}))


if __name__ == "__main__":
    main()
