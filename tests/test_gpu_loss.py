"""GPU parity of the fused L1+SSIM loss (gsrast.losses.l1_ssim -> gsr_loss_l1_ssim) vs the oracle and, at 1080p, the torch chain."""
import numpy as np
import pytest
import torch

import oracle
import ref_loss_torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(shape, seed, noise=0.15):
    r = np.random.default_rng(seed)
    gt = r.uniform(0, 1, shape).astype(np.float32)
    return np.clip(gt + r.normal(0, noise, shape), 0, 1).astype(np.float32), gt


@pytest.mark.parametrize("shape,lam", [((3, 37, 53), 0.2), ((1, 8, 9), 0.5), ((3, 64, 48), 1.0), ((2, 11, 30), 0.0), ((3, 16, 16), 0.2),
                                       ((3, 1, 1), 0.2), ((3, 200, 333), 0.2)])
def test_l1_ssim_matches_oracle(shape, lam):
    from gsrast.losses import l1_ssim
    img, gt = _pair(shape, sum(shape))
    out, d = oracle.loss_l1_ssim(img, gt, lam)
    x = torch.tensor(img, device=DEV, requires_grad=True)
    loss, parts = l1_ssim(x, torch.tensor(gt, device=DEV), lam, return_parts=True)
    (loss * 3.0).backward()
    np.testing.assert_allclose([parts[0].item(), parts[1].item(), loss.item()], out, rtol=2e-5, atol=2e-6)
    g = x.grad.cpu().numpy() / 3.0
    assert np.abs(g - d).max() <= 1e-4 * np.abs(d).max() + 1e-9


def test_l1_ssim_full_hd_vs_torch_chain():
    from gsrast.losses import l1_ssim
    img, gt = _pair((3, 1080, 1920), 5)
    x = torch.tensor(img, device=DEV, requires_grad=True)
    y = torch.tensor(gt, device=DEV)
    loss = l1_ssim(x, y, 0.2)
    loss.backward()
    xr = torch.tensor(img, device=DEV, requires_grad=True)
    Lr, _, _ = ref_loss_torch.loss(xr.unsqueeze(0), y.unsqueeze(0), 0.2)
    Lr.backward()
    assert abs(loss.item() - Lr.item()) < 1e-5
    g, gr = x.grad, xr.grad
    assert (g - gr).abs().max().item() <= 2e-3 * gr.abs().max().item()
    assert ((g - gr).norm() / gr.norm()).item() < 1e-4
    # identical images: SSIM = 1, loss = 0
    l0, parts = l1_ssim(y, y, 0.2, return_parts=True)
    assert abs(parts[1].item() - 1.0) < 1e-6 and abs(l0.item()) < 1e-6


def _geo_inputs(H, W, seed):
    import test_loss_cpu
    import ref_geo_torch
    am, cam = test_loss_cpu._geo_case(H, W, seed)
    wvt = torch.tensor(cam["viewmatrix"]); fpt = torch.tensor(cam["projmatrix"])
    return am, wvt, fpt


@pytest.mark.parametrize("H,W,ratio,seed", [(23, 31, 0.0, 0), (17, 40, 1.0, 1), (30, 22, 0.3, 2), (3, 3, 0.0, 3), (2, 5, 0.0, 4), (16, 16, 0.0, 5),
                                            (33, 49, 0.0, 6), (200, 333, 0.0, 7)])
def test_surfel_geo_matches_oracle(H, W, ratio, seed):
    from gsrast.losses import camera_ray_matrices, surfel_geo_loss
    am, wvt, fpt = _geo_inputs(H, W, seed)
    rm, nr = camera_ray_matrices(wvt.to(DEV), fpt.to(DEV), W, H)
    o = oracle.loss_surfel_geo(am, rm.cpu().numpy(), nr.cpu().numpy(), ratio, 0.05, 100.0)
    x = torch.tensor(am, device=DEV, requires_grad=True)
    loss, parts, depth, nw, sn = surfel_geo_loss(x, rm, nr, ratio, 0.05, 100.0, return_maps=True)
    (2.0 * loss).backward()
    np.testing.assert_allclose([parts[0].item(), parts[1].item(), loss.item()], o["loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(depth.cpu().numpy(), o["surf_depth"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(nw.cpu().numpy(), o["normal_world"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(sn.cpu().numpy(), o["surf_normal"], rtol=0, atol=2e-4)     # normalize of a near-degenerate cross product
    g = x.grad.cpu().numpy() / 2.0
    r = o["dL_dallmap"]
    assert np.abs(g - r).max() <= 2e-3 * np.abs(r).max() + 1e-12
    assert np.linalg.norm((g - r).ravel()) / (np.linalg.norm(r.ravel()) + 1e-30) < 1e-4


def test_surfel_geo_full_hd_vs_torch_chain():
    import ref_geo_torch
    from gsrast.losses import camera_ray_matrices, surfel_geo_loss
    H, W = 1080, 1920
    am, wvt, fpt = _geo_inputs(H, W, 9)
    wvt, fpt = wvt.to(DEV), fpt.to(DEV)
    rm, nr = camera_ray_matrices(wvt, fpt, W, H)
    x = torch.tensor(am, device=DEV, requires_grad=True)
    loss, parts = surfel_geo_loss(x, rm, nr, 0.0, 0.05, 100.0)
    loss.backward()
    xr = torch.tensor(am, device=DEV, requires_grad=True)
    Lr, ne, dm, post = ref_geo_torch.geo_loss(xr, wvt, fpt, 0.0, 0.05, 100.0)
    Lr.backward()
    assert abs(loss.item() - Lr.item()) < 2e-5 * abs(Lr.item()) + 1e-6
    g, gr = x.grad, torch.nan_to_num(xr.grad, 0.0, 0.0, 0.0)
    assert ((g - gr).norm() / gr.norm()).item() < 2e-3        # fp32 chain vs fp32 fused: cancellation in the cross products


@pytest.mark.parametrize("H,W,seed,use_w", [(23, 31, 0, True), (17, 40, 1, False), (3, 3, 2, True), (2, 6, 3, True), (100, 161, 4, True)])
def test_plane_geo_matches_oracle(H, W, seed, use_w):
    import test_loss_cpu
    from gsrast.losses import plane_geo_loss
    depth, am, K, weight = test_loss_cpu._plane_case(H, W, seed)
    w = weight if use_w else None
    rm = torch.inverse(torch.tensor(K, dtype=torch.float64).t()).float()
    o = oracle.loss_plane_geo(depth, am[3], am[0:3], w, rm.numpy(), 0.015)
    d = torch.tensor(depth, device=DEV).unsqueeze(0).requires_grad_(True)
    a = torch.tensor(am, device=DEV, requires_grad=True)
    loss, part, dn = plane_geo_loss(d, a, rm.to(DEV), None if w is None else torch.tensor(w, device=DEV), 0.015, return_map=True)
    (3.0 * loss).backward()
    np.testing.assert_allclose([part[0].item(), loss.item()], o["loss"][[0, 2]], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(dn.cpu().numpy(), o["depth_normal"], rtol=0, atol=2e-4)
    gd = d.grad.cpu().numpy()[0] / 3.0
    assert np.abs(gd - o["dL_ddepth"]).max() <= 2e-3 * np.abs(o["dL_ddepth"]).max() + 1e-12
    ga = a.grad.cpu().numpy() / 3.0
    flips = np.sign(ga[0:3]) != np.sign(o["dL_dnormal"])           # sign(depth_normal - normal) can flip where the difference is ~0
    assert flips.mean() < 1e-3 and not ga[3:].any()
    assert np.abs(np.abs(ga[0:3]) - np.abs(o["dL_dnormal"])).max() <= 1e-6 * np.abs(o["dL_dnormal"]).max() + 1e-12


def test_plane_geo_full_hd_vs_torch_chain():
    import ref_geo_torch
    import test_loss_cpu
    from gsrast.losses import plane_geo_loss
    H, W = 1080, 1920
    depth, am, K, weight = test_loss_cpu._plane_case(H, W, 8)
    Kt = torch.tensor(K, device=DEV)
    rm = torch.inverse(Kt.double().t()).float()
    d = torch.tensor(depth, device=DEV).requires_grad_(True); a = torch.tensor(am, device=DEV, requires_grad=True)
    wt = torch.tensor(weight, device=DEV)
    loss, part = plane_geo_loss(d, a, rm, wt, 0.015)
    loss.backward()
    dr = torch.tensor(depth, device=DEV).requires_grad_(True); ar = torch.tensor(am, device=DEV, requires_grad=True)
    Lr, m, dn = ref_geo_torch.plane_geo_loss(dr, ar, Kt, wt, 0.015)
    Lr.backward()
    assert abs(loss.item() - Lr.item()) < 2e-5 * abs(Lr.item()) + 1e-7
    assert ((d.grad - dr.grad).norm() / dr.grad.norm()).item() < 5e-3


def test_l1_plus_linear_root_flag_same_gradients():
    from gsrast.losses import l1_plus_linear
    g = torch.Generator().manual_seed(0)
    c = torch.rand(3, 40, 56, generator=g).to(DEV); gt = torch.rand(3, 40, 56, generator=g).to(DEV)
    a = torch.rand(11, 40, 56, generator=g).to(DEV); w = torch.randn(11, 40, 56, generator=g).to(DEV)
    grads = []
    for root in (False, True):
        cc = c.clone().requires_grad_(True); aa = a.clone().requires_grad_(True)
        loss = l1_plus_linear(cc, gt, aa, w, root=root)
        loss.backward()
        grads.append((loss.item(), cc.grad.clone(), aa.grad.clone()))
    assert abs(grads[0][0] - grads[1][0]) < 1e-4 * abs(grads[0][0])          # the value is summed with float atomics: order-dependent
    assert torch.equal(grads[0][1], grads[1][1]) and torch.equal(grads[0][2], grads[1][2])
    ref = (c - gt).abs().mean() + (a * w).sum()
    assert abs(grads[0][0] - ref.item()) < 1e-3 * abs(ref.item()) + 1e-5


@pytest.mark.parametrize("shape, shift", [((540, 960), 0), ((67, 91), 0), ((67, 91), 1)], ids=["several-grid-strides", "ragged", "misaligned"])
def test_l1_plus_linear_matches_torch(shape, shift):
    """Value and dL/dcolor against the torch expression, at a size that takes the unrolled grid-stride loops through whole rounds and a
    remainder (11 x 540 x 960 auxiliary values: 2.7 rounds of the 2048 x 256-thread grid), at a ragged size (tail elements after the float4
    body), and on views that start one element into their storage (the 16-byte test fails: scalar path)."""
    from gsrast.losses import l1_plus_linear
    H, W = shape
    g = torch.Generator().manual_seed(3)
    def mk(ch, randn=False):
        t = (torch.randn if randn else torch.rand)(ch * H * W + shift, generator=g).to(DEV)
        return t[shift:].view(ch, H, W)
    c, gt, a, w = mk(3), mk(3), mk(11), mk(11, True)
    cc = c.detach().requires_grad_(True)
    loss = l1_plus_linear(cc, gt, a, w)
    loss.backward()
    ref_c = c.detach().double().requires_grad_(True)
    ref = (ref_c - gt.double()).abs().mean() + (a.double() * w.double()).sum()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-5 * max(abs(ref.item()), (a * w).abs().sum().item() * 1e-2) + 1e-6
    assert torch.equal(cc.grad, ref_c.grad.float())


def test_unit_upstream_flag_same_gradients_and_unused_loss_sends_none():
    """unit_upstream=True (the loss enters the total with weight 1, as in GS-SR's sum(loss_dict.values())) returns the stored gradient without the
    full-size multiply by the upstream scalar: bit-identical to the default path when that scalar is 1; and a loss value that is computed but
    never used sends no gradient at all."""
    from gsrast.losses import camera_ray_matrices, l1_ssim, plane_geo_loss, surfel_geo_loss
    g = torch.Generator().manual_seed(1)
    H, W = 48, 80
    img = torch.rand(3, H, W, generator=g).to(DEV); gt = torch.rand(3, H, W, generator=g).to(DEV)
    am = torch.rand(11, H, W, generator=g).to(DEV) + 0.2
    oam = torch.rand(5, H, W, generator=g).to(DEV) + 0.1; pd = torch.rand(1, H, W, generator=g).to(DEV) + 1.0
    rm = torch.eye(3, device=DEV) + 0.01 * torch.rand(3, 3, generator=g).to(DEV); nr = torch.eye(3, device=DEV)
    out = []
    for unit in (False, True):
        a, b, c, d = (t.clone().requires_grad_(True) for t in (img, am, oam, pd))
        total = (l1_ssim(a, gt, 0.2, unit_upstream=unit) + surfel_geo_loss(b, rm, nr, 0.0, 0.05, 100.0, unit_upstream=unit)[0]
                 + plane_geo_loss(d, c, rm, None, 0.015, unit_upstream=unit)[0])
        total.backward()
        out.append([t.grad.clone() for t in (a, b, c, d)])
    for x, y in zip(*out):
        assert torch.equal(x, y) and x.abs().max() > 0
    a = img.clone().requires_grad_(True); b = am.clone().requires_grad_(True)
    unused = surfel_geo_loss(b, rm, nr, 0.0, 0.05, 100.0)[0]          # computed, not part of the total
    l1_ssim(a, gt, 0.2).backward()
    assert b.grad is None and a.grad is not None and unused.item() == unused.item()


@pytest.mark.parametrize("cols,width", [(3, 3), (2, 3), (2, 2), (1, 1)])
def test_scaling_prod_mean_matches_torch(cols, width):
    """scaling_loss of the scaffold / octree scenes (scaffold_scene.py:184: lambda * scaling.prod(dim=1).mean()) against the torch expression,
    value and gradient, incl. rows with a zero scale (where torch's prod backward takes its special path), a device-side divisor, and an
    upstream factor."""
    from gsrast.losses import scaling_prod_mean
    g = torch.Generator().manual_seed(4)
    P = 70001
    s0 = torch.rand(P, width, generator=g) * 0.1 + 0.001
    s0[17] = 0.0; s0[99, 0] = 0.0
    a = s0.to("cuda").requires_grad_(True); b = s0.to("cuda").requires_grad_(True)
    la = scaling_prod_mean(a, 0.01, cols=cols)
    lb = 0.01 * b[:, :cols].prod(dim=1).mean()
    (3.0 * la).backward(); (3.0 * lb).backward()
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb)) + 1e-12
    assert torch.allclose(a.grad, b.grad, rtol=2e-6, atol=1e-12)
    assert cols == width or float(a.grad[:, cols:].abs().max()) == 0.0
    # static-shape form: trailing rows parked at scale 0, divisor from the device
    cnt = torch.tensor([P - 5000], dtype=torch.int32, device="cuda")
    s1 = s0.clone(); s1[P - 5000:] = 0.0
    c = s1.to("cuda").requires_grad_(True); d = s1.to("cuda").requires_grad_(True)
    lc = scaling_prod_mean(c, 0.01, cols=cols, count=cnt, unit_upstream=True)
    ld = 0.01 * d[:P - 5000, :cols].prod(dim=1).mean()
    lc.backward(); ld.backward()
    assert abs(float(lc) - float(ld)) <= 1e-6 * abs(float(ld)) + 1e-12
    assert torch.allclose(c.grad, d.grad, rtol=2e-6, atol=1e-12)
