"""GPU test (pytest -m gpu) of the drop-in as a TRAINING COMPONENT -- SURVEY.md §8 row (a)24, the callers' contract beyond one call.

Per variant: a 20 000-Gaussian teacher scene is rendered once through the drop-in package; its parameters are perturbed (means, log-scales, quaternions,
opacity logits, colours) and optimised back for 300 iterations of exactly the loop GS-SR runs (gssr/engine/trainer.py:118-130):
    activations (vanilla_gaussian.py:250-269) -> GaussianRasterizer -> L1 + D-SSIM (vanilla_scene.py:63-69; + a small geometric term on the 2DGS / PGSR
    maps so that every gradient input of the backward is live) -> loss.backward() -> densification statistics from `viewspace_points.grad[:, :2]`,
    `radii > 0` and (PGSR) `out_observe` / `viewspace_points_abs` (vanilla_gaussian.py:467-472,428-430; pgsr_gaussian.py:157-172) -> Adam with the
    reference's per-group learning rates (vanilla_gaussian.py:120-139: `torch.optim.Adam(l, lr=0.0, eps=1e-15)`, here gsrast.optim.Adam).
Asserted: PSNR against the teacher image rises by >= 6 dB; the loss never rises by more than 5 % over any 20-step window; the accumulated statistics equal a
float64 recomputation from the per-iteration gradients / radii / observe counts saved on the way."""
import math

import numpy as np
import pytest
import torch

import scenes

pytestmark = pytest.mark.gpu

W, H, P, ITERS = 640, 360, 20000, 300


def _psnr(a, b):
    return -10.0 * math.log10(float(((a - b) ** 2).mean()))


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_a_perturbed_scene_trains_back_to_its_teacher_image(variant):
    import diff_gaussian_rasterization as dgr
    import diff_plane_rasterization as dpr
    import diff_surfel_rasterization as dsr
    from gsrast import runner
    from gsrast.activations import gaussian_activations
    from gsrast.losses import l1_ssim
    from gsrast.optim import Adam
    from gsrast.plane_prep import plane_input_all_map
    from gsrast.stats import densification_stats_

    sc = scenes.make_scene(variant, P, W, H, seed=5, sigma_px=6.0, bg=(0.0, 0.0, 0.0))
    t = runner.to_dev(sc, "cuda")
    rs = runner.settings(variant, t)
    dev = t["means3D"].device

    def render(xyz, scaling_raw, rot_raw, opacity_raw, col, means2D, means2D_abs=None):
        s, q, o = gaussian_activations(scaling_raw, rot_raw, opacity_raw)
        kw = dict(means3D=xyz, means2D=means2D, opacities=o, colors_precomp=col, scales=s, rotations=q)
        if variant == "surfel":
            color, radii, allmap = dsr.GaussianRasterizer(rs)(**kw)
            return color, radii, None, 1.0 * allmap[6].mean()                                   # depth distortion (twodgs_scene.py:99-105 weights it with lambda_dist)
        if variant == "plane":
            am = plane_input_all_map(xyz, q, s, t["viewmatrix"], t["campos"])                   # pgsr_scene.py:297-304
            color, radii, observe, out_all_map, plane_depth = dpr.GaussianRasterizer(rs)(means2D_abs=means2D_abs, all_map=am, **kw)
            return color, radii, observe, 0.01 * (1.0 - out_all_map[3]).mean() + 0.001 * plane_depth.mean()
        color, radii = dgr.GaussianRasterizer(rs)(**kw)
        return color, radii, None, None

    g = torch.Generator().manual_seed(0)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)
    zero2d = lambda: torch.zeros((P, 3), device=dev, requires_grad=True)
    true = dict(xyz=t["means3D"], scaling=torch.log(t["scales"]), rot=t["rotations"], opacity=torch.logit(t["opacities"].clamp(1e-4, 1 - 1e-4)), col=t["colors_precomp"])
    with torch.no_grad():
        teacher = render(true["xyz"], true["scaling"], true["rot"], true["opacity"], true["col"], zero2d(), zero2d())[0].clone()
    depth = (t["means3D"] @ t["viewmatrix"][:3, 2] + t["viewmatrix"][3, 2]).abs().clamp_min(1.0)[:, None]
    prm = dict(xyz=true["xyz"] + 0.004 * depth * rnd(P, 3), scaling=true["scaling"] + 0.25 * rnd(*true["scaling"].shape), rot=true["rot"] + 0.05 * rnd(P, 4),
               opacity=true["opacity"] + 0.7 * rnd(P, 1), col=true["col"] + 0.2 * rnd(P, 3))
    prm = {k: torch.nn.Parameter(v.contiguous()) for k, v in prm.items()}
    spatial_lr_scale = 5.0
    opt = Adam([{"params": [prm["xyz"]], "lr": 0.00016 * spatial_lr_scale, "name": "xyz"}, {"params": [prm["col"]], "lr": 0.0025, "name": "f_dc"},
                {"params": [prm["opacity"]], "lr": 0.05, "name": "opacity"}, {"params": [prm["scaling"]], "lr": 0.005, "name": "scaling"},
                {"params": [prm["rot"]], "lr": 0.001, "name": "rotation"}], lr=0.0, eps=1e-15)
    z = lambda: torch.zeros(P, device=dev)
    max_radii2D, accum, denom, accum_abs, denom_abs = z(), z(), z(), z(), z()
    saved, losses, psnr = [], [], []
    for it in range(ITERS):
        means2D, means2D_abs = zero2d(), (zero2d() if variant == "plane" else None)            # screenspace_points: only their .grad is used
        color, radii, observe, geo = render(prm["xyz"], prm["scaling"], prm["rot"], prm["opacity"], prm["col"], means2D, means2D_abs)
        loss = l1_ssim(color, teacher, 0.2)
        if geo is not None:
            loss = loss + geo
        loss.backward()
        vis = radii > 0
        if variant == "plane":
            densification_stats_(max_radii2D, accum, denom, means2D.grad, vis, radii, out_observe=observe, viewspace_grad_abs=means2D_abs.grad,
                                 xyz_gradient_accum_abs=accum_abs, denom_abs=denom_abs)
        else:
            densification_stats_(max_radii2D, accum, denom, means2D.grad, vis, radii)
        saved.append((means2D.grad[:, :2].double().cpu().numpy(), radii.cpu().numpy(), None if observe is None else observe.cpu().numpy(),
                      None if means2D_abs is None else means2D_abs.grad[:, :2].double().cpu().numpy()))
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(loss)); psnr.append(_psnr(color.detach(), teacher))
    # ---- it trains
    L = np.asarray(losses)
    worst = float((L[20:] / L[:-20]).max())
    print(f"convergence[{variant}]: PSNR {psnr[0]:.2f} -> {psnr[-1]:.2f} dB, loss {L[0]:.4f} -> {L[-1]:.4f}, worst 20-step ratio {worst:.3f}")
    assert psnr[-1] - psnr[0] >= 6.0, (psnr[0], psnr[-1])
    assert worst <= 1.05, (worst, int((L[20:] / L[:-20]).argmax()))
    assert L[-1] < 0.6 * L[0]
    # ---- and the statistics the densification reads are what the saved per-iteration gradients say (float64)
    r_acc, r_den, r_max, r_acc_abs = np.zeros(P), np.zeros(P), np.zeros(P), np.zeros(P)
    for g2, rad, obs, g2a in saved:
        v = rad > 0
        r_acc[v] += np.linalg.norm(g2[v], axis=-1); r_den[v] += 1
        m = v & (obs > 0) if obs is not None else v
        r_max[m] = np.maximum(r_max[m], rad[m])
        if g2a is not None:
            r_acc_abs[v] += np.linalg.norm(g2a[v], axis=-1)
    assert np.array_equal(denom.cpu().numpy(), r_den) and np.array_equal(max_radii2D.cpu().numpy(), r_max)
    assert r_den.max() == ITERS and (r_acc > 0).mean() > 0.5
    np.testing.assert_allclose(accum.cpu().numpy(), r_acc, rtol=2e-5, atol=1e-12)
    if variant == "plane":
        assert np.array_equal(denom_abs.cpu().numpy(), r_den) and (r_acc_abs >= r_acc * (1 - 1e-6)).all()
        np.testing.assert_allclose(accum_abs.cpu().numpy(), r_acc_abs, rtol=2e-5, atol=1e-12)
