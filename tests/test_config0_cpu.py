"""BASELINE.json configs[0]: "vanilla-3dgs, 1 iter CPU raster (plumbing, no GPU)".

The reference cannot rasterize on a CPU; the CPU oracle can.  This test drives it through a minimal caller that builds
exactly the kwargs VanillaScene.render passes (gssr/scene/vanilla_scene.py:112-159: shs (P,16,3) with active_sh_degree 0 at iteration 1,
activated opacity / scaling / rotation, zero screenspace points with retained grad) and runs a few training iterations
(L1 loss -> backward -> Adam with the per-group learning rates of gssr/gaussian/vanilla_gaussian.py) entirely on CPU tensors.
TEST INFRASTRUCTURE: the oracle is the rasterizer here because no GPU exists on this box; the product has no CPU path."""
import numpy as np
import torch

import oracle
import scenes


class _OracleRaster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sc, means3D, means2D, shs, opacity, scales, rotations):
        s = dict(sc)
        s.update(means3D=means3D.detach().numpy(), shs=shs.detach().numpy(), opacities=opacity.detach().numpy().reshape(-1),
                 scales=scales.detach().numpy(), rotations=rotations.detach().numpy(), colors_precomp=None)
        f = oracle.Forward(s, "ewa")
        ctx.f = f
        return torch.from_numpy(f.color.copy()), torch.from_numpy(f.radii.copy())

    @staticmethod
    def backward(ctx, g_color, _):
        g = ctx.f.backward(dL_dcolor=g_color.contiguous().numpy())
        ctx.f.close()
        t = torch.from_numpy
        return (None, t(g["dL_dmeans3D"]), t(g["dL_dmeans2D"]), t(g["dL_dsh"]), t(g["dL_dopacity"]), t(g["dL_dscales"]), t(g["dL_drotations"]))


def test_vanilla_3dgs_training_iterations_on_cpu():
    W, H, P = 96, 64, 400
    sc = scenes.make_scene("ewa", P, W, H, seed=3, color_mode="sh", sh_degree=0)          # active_sh_degree = 0 at iteration 1
    assert sc["shs"].shape == (P, 16, 3) and sc["sh_degree"] == 0
    # trainable parameters in the reference's raw parametrisation
    xyz = torch.tensor(sc["means3D"], requires_grad=True)
    f_dc = torch.tensor(sc["shs"][:, :1], requires_grad=True); f_rest = torch.tensor(sc["shs"][:, 1:], requires_grad=True)
    opa_raw = torch.logit(torch.tensor(sc["opacities"]).clamp(1e-4, 1 - 1e-4)).reshape(-1, 1).requires_grad_(True)
    sca_raw = torch.log(torch.tensor(sc["scales"])).requires_grad_(True)
    rot_raw = torch.tensor(sc["rotations"], requires_grad=True)
    opt = torch.optim.Adam([{"params": [xyz], "lr": 1.6e-4}, {"params": [f_dc], "lr": 2.5e-3}, {"params": [f_rest], "lr": 2.5e-3 / 20},
                            {"params": [opa_raw], "lr": 0.05}, {"params": [sca_raw], "lr": 5e-3}, {"params": [rot_raw], "lr": 1e-3}], eps=1e-15)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(0))
    losses = []
    for it in range(4):
        screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0
        screenspace_points.retain_grad()
        shs = torch.cat([f_dc, f_rest], dim=1)
        color, radii = _OracleRaster.apply(sc, xyz, screenspace_points, shs, torch.sigmoid(opa_raw), torch.exp(sca_raw),
                                           torch.nn.functional.normalize(rot_raw))
        out = {"render": color, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
        loss = torch.abs(out["render"] - gt).mean()
        loss.backward()
        assert out["viewspace_points"].grad is not None and out["viewspace_points"].grad.shape == (P, 3)
        assert out["visibility_filter"].sum() > 0.5 * P
        assert f_rest.grad is not None and not f_rest.grad.any()            # degree 0: higher bands get exactly zero gradient
        opt.step(); opt.zero_grad(set_to_none=True)
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
