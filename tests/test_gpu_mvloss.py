"""GPU parity of the PGSR multi-view losses (gsrast.losses.plane_multiview_loss -> gsr_loss_plane_mv_{geo,ncc}) against the C oracle
(oracle/gsm_oracle.c, itself pinned to a reference run) and the reference-run fixture."""
import numpy as np
import pytest
import torch

import golden_ref
import mv_cases
import oracle_multiview as om

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run_hip(case, indices=None, lam_geo=0.03, lam_ncc=0.15, th=1.0, patch=3, num_sample=102400, gen=None):
    from gsrast.losses import multiview_cfg, plane_multiview_loss
    cfg = multiview_cfg(mv_cases.cam_ns(case["view"]), mv_cases.cam_ns(case["near"]), case["W"], case["H"], near_size=(case["W"], case["H"]),
                        patch_size=patch, pixel_noise_threshold=th)
    t = lambda a: torch.tensor(a, device=DEV)
    leaves = {k: t(case[k]).requires_grad_(True) for k in ("plane_depth", "near_plane_depth", "rendered_normal", "rendered_distance")}
    geo, ncc, aux = plane_multiview_loss(leaves["plane_depth"], leaves["near_plane_depth"], leaves["rendered_normal"], leaves["rendered_distance"],
                                         t(case["gray"]), t(case["near_gray"]), cfg, lam_geo, lam_ncc, num_sample=num_sample,
                                         indices=None if indices is None else t(indices), generator=gen, return_aux=True)
    (2.0 * geo + 3.0 * ncc).backward()
    g = {k: v.grad.cpu().numpy() for k, v in leaves.items()}
    return geo.item(), ncc.item(), {k: v.cpu().numpy() for k, v in aux.items()}, g


def _rel(a, b):
    return np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30)


@pytest.mark.parametrize("kw", [dict(), dict(W=200, H=120, seed=1, amp=0.05, tex=3.0), dict(W=33, H=17, seed=2), dict(W=96, H=64, seed=3, amp=0.3, near_yaw=-12.0),
                                dict(W=64, H=48, seed=4, near_t=(-3.0, 0.0, 0.0))])
def test_multiview_matches_oracle(kw):
    case = mv_cases.plane_pair(**kw)
    W, H = case["W"], case["H"]
    cfg = om.make_cfg(W, H, case["view"], case["near"])
    og = om.geo(cfg, case["plane_depth"], case["near_plane_depth"])
    geo, ncc, aux, g = _run_hip(case)
    # the gates (frustum, noise < th) are float comparisons: allow a vanishing number of flips, compare the rest exactly
    flips = aux["d_mask"].reshape(-1) != og["dmask"].astype(bool)
    assert flips.sum() <= max(2, int(2e-4 * W * H)), int(flips.sum())
    ok = ~flips
    np.testing.assert_allclose(aux["pixel_noise"].reshape(-1)[ok & og["dmask"].astype(bool)], og["noise"][ok & og["dmask"].astype(bool)], rtol=0, atol=2e-5)
    cnt = og["stats"][1]
    if cnt == 0:
        assert geo == 0.0 and ncc == 0.0 and not any(np.abs(v).max() for v in g.values())
        return
    if not flips.any():
        np.testing.assert_allclose(geo, 0.03 * og["stats"][0] / cnt, rtol=1e-4)
        assert _rel(g["plane_depth"].reshape(-1) / 2.0, 0.03 / cnt * og["g_depth"]) < 1e-3
        assert _rel(g["near_plane_depth"].reshape(-1) / 2.0, 0.03 / cnt * og["g_near"]) < 1e-3
    # NCC on the oracle's own sample set and weights, so that it is compared independently of the geo gates
    idx = np.nonzero(og["dmask"])[0].astype(np.int32)
    on = om.ncc(cfg, idx, og["weight"], case["rendered_normal"], case["rendered_distance"], case["gray"], case["near_gray"])
    geo2, ncc2, aux2, g2 = _run_hip(case, indices=idx)
    mflip = aux2["ncc_mask"] != on["mask"].astype(bool)
    assert mflip.sum() <= max(2, int(1e-3 * idx.size))
    # Per-tap arithmetic follows the oracle's operation order (true divisions, same bilinear accumulation, -ffp-contract=off); the patch
    # sums are 16-lane butterflies instead of a sequential walk.  ncc = 1 - cross^2/(var var) is ill-conditioned on low-contrast patches
    # (the variances cancel to ~1e-3 of the float32 sums), where the summation order alone moves it by ~1e-3: bulk tight, tail loose.
    e = np.abs(aux2["ncc"][~mflip] - on["ncc"][~mflip])
    assert np.quantile(e, 0.5) < 2e-5 and np.quantile(e, 0.99) < 2e-3 and e.max() < 3e-2, (np.quantile(e, 0.5), np.quantile(e, 0.99), e.max())
    if on["stats"][1] > 0 and mflip.sum() <= 2 and not flips.any():
        np.testing.assert_allclose(ncc2, 0.15 * on["stats"][0] / on["stats"][1], rtol=1e-4)
        sc = 0.15 / on["stats"][1]
        assert _rel(g2["rendered_normal"].reshape(3, -1) / 3.0, sc * on["g_normal"]) < 5e-3       # conditioning as above
        assert _rel(g2["rendered_distance"].reshape(-1) / 3.0, sc * on["g_dist"]) < 5e-3


def test_multiview_matches_reference_run():
    z = golden_ref.load("ref_loss_plane_multiview")
    cam = lambda pre: {k: (z[f"{pre}_{k}"] if k in ("R", "T") else float(z[f"{pre}_{k}"])) for k in ("R", "T", "Fx", "Fy", "Cx", "Cy")}
    case = dict(W=int(z["W"]), H=int(z["H"]), view=cam("v"), near=cam("n"), **{k: z[k] for k in ("plane_depth", "near_plane_depth", "rendered_normal",
                                                                                                 "rendered_distance", "gray", "near_gray")})
    geo, ncc, aux, g = _run_hip(case, lam_geo=float(z["lambda_geo"]), lam_ncc=float(z["lambda_ncc"]), th=float(z["pixel_noise_threshold"]),
                                patch=int(z["patch_size"]))
    np.testing.assert_allclose(geo, float(z["geo_loss"]), rtol=1e-4)
    np.testing.assert_allclose(ncc, float(z["ncc_loss"]), rtol=1e-3)
    assert _rel(g["plane_depth"] / 2.0, z["d_plane_depth"]) < 1e-3
    assert _rel(g["near_plane_depth"] / 2.0, z["d_near_plane_depth"]) < 1e-3
    assert _rel(g["rendered_normal"] / 3.0, z["d_rendered_normal"]) < 1e-3
    assert _rel(g["rendered_distance"] / 3.0, z["d_rendered_distance"]) < 1e-3


def test_multiview_sampling_and_empty_masks():
    from gsrast.losses import sample_valid_pixels
    case = mv_cases.plane_pair(W=160, H=120, seed=7, amp=0.05)
    gen = torch.Generator(device=DEV); gen.manual_seed(3)
    geo, ncc, aux, g = _run_hip(case, num_sample=2000, gen=gen)
    idx = aux["indices"]
    assert idx.size == 2000 and (idx >= 0).all() and np.unique(idx).size == 2000 and aux["d_mask"].reshape(-1)[idx].all()
    # the sampler itself: exact count, no repeats, only valid pixels, ascending, deterministic in the seed, roughly uniform over the mask
    dm = torch.tensor(aux["d_mask"].reshape(-1), device=DEV)
    s1 = sample_valid_pixels(dm, 3000, seed=11).cpu().numpy(); s1b = sample_valid_pixels(dm, 3000, seed=11).cpu().numpy()
    s2 = sample_valid_pixels(dm, 3000, seed=12).cpu().numpy()
    assert np.array_equal(s1, s1b) and not np.array_equal(s1, s2)
    for sset in (s1, s2):
        assert (sset >= 0).all() and np.unique(sset).size == 3000 and aux["d_mask"].reshape(-1)[sset].all()
    assert (np.diff(s1[:2990]) > 0).all()                                   # ascending (the last few slots may hold threshold ties)
    valid = np.nonzero(aux["d_mask"].reshape(-1))[0]
    pooled = np.concatenate([sample_valid_pixels(dm, 3000, seed=100 + k).cpu().numpy() for k in range(20)])
    rank = np.searchsorted(valid, pooled) / valid.size                      # position of each draw within the valid set: ~U(0,1)
    hist = np.histogram(rank, bins=10, range=(0, 1))[0]
    assert np.abs(hist / hist.sum() - 0.1).max() < 0.01, hist
    counts = np.bincount(np.searchsorted(valid, pooled), minlength=valid.size)
    assert counts.max() <= 20 and abs(counts.mean() - 20 * 3000 / valid.size) < 1e-9
    assert ncc > 0 and (np.abs(g["rendered_distance"]).reshape(-1) > 0).sum() <= 2000
    # fewer valid pixels than slots: every valid pixel exactly once, the rest -1
    m = torch.zeros(50, dtype=torch.bool, device=DEV); m[[3, 7, 11]] = True
    big = torch.zeros(5000, dtype=torch.bool, device=DEV); big[[5, 4000]] = True
    assert sorted(sample_valid_pixels(m, 100).cpu().tolist()) == sorted([-1] * 47 + [3, 7, 11])
    s = sample_valid_pixels(big, 100).cpu().numpy()
    assert sorted(s[s >= 0].tolist()) == [5, 4000] and s.size == 100
    # neighbour camera looking away: empty d_mask -> both losses 0, all gradients 0, no NaN
    far = mv_cases.plane_pair(W=64, H=48, seed=8, near_yaw=170.0)
    geo, ncc, aux, g = _run_hip(far)
    assert geo == 0.0 and ncc == 0.0 and not aux["d_mask"].any()
    assert all(np.isfinite(v).all() and not v.any() for v in g.values())


def test_multiview_full_hd_vs_torch_chain():
    """1920x1080, 102400 sampled patches: fused kernels vs the torch op chain on the same device and the same sample set."""
    import ref_mv_torch
    from gsrast.losses import multiview_cfg, plane_multiview_loss
    case = mv_cases.plane_pair(W=1920, H=1080, seed=11, amp=0.002, tex=25.0)
    t = lambda a: torch.tensor(a, device=DEV)
    names = ("plane_depth", "near_plane_depth", "rendered_normal", "rendered_distance")
    a = {k: t(case[k]).requires_grad_(True) for k in names}
    b = {k: t(case[k]).requires_grad_(True) for k in names}
    cfg = multiview_cfg(mv_cases.cam_ns(case["view"]), mv_cases.cam_ns(case["near"]), 1920, 1080, near_size=(1920, 1080))
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    geo, ncc, aux = plane_multiview_loss(*[a[k] for k in names], t(case["gray"]), t(case["near_gray"]), cfg, generator=gen, return_aux=True)
    (geo + ncc).backward()
    idx = aux["indices"]
    assert idx.numel() == 102400 and (idx >= 0).all()
    rgeo, rncc = ref_mv_torch.multiview_loss(*[b[k] for k in names], t(case["gray"]), t(case["near_gray"]), case["view"], case["near"], indices=idx)
    (rgeo + rncc).backward()
    assert abs(geo.item() - rgeo.item()) < 2e-3 * abs(rgeo.item()) and abs(ncc.item() - rncc.item()) < 2e-3 * abs(rncc.item())
    for k in names:
        rel = ((a[k].grad - b[k].grad).norm() / b[k].grad.norm()).item()
        assert rel < 2e-2, (k, rel)          # float32 chain through world coordinates vs composed transforms; NCC conditioning (see above)


@pytest.mark.parametrize("patch", [1, 4])
def test_multiview_other_patch_sizes(patch):
    """patch = 4 (81 taps) takes the re-gathering kernel variant, patch = 1 the register-cached one with idle lanes."""
    case = mv_cases.plane_pair(W=96, H=64, seed=9, tex=2.0)
    cfg = om.make_cfg(96, 64, case["view"], case["near"], patch=patch)
    og = om.geo(cfg, case["plane_depth"], case["near_plane_depth"])
    idx = np.nonzero(og["dmask"])[0].astype(np.int32)
    on = om.ncc(cfg, idx, og["weight"], case["rendered_normal"], case["rendered_distance"], case["gray"], case["near_gray"])
    _, ncc, aux, g = _run_hip(case, indices=idx, patch=patch)
    mflip = aux["ncc_mask"] != on["mask"].astype(bool)
    assert mflip.sum() <= max(2, int(1e-3 * idx.size))
    e = np.abs(aux["ncc"][~mflip] - on["ncc"][~mflip])
    assert np.quantile(e, 0.5) < 2e-5 and np.quantile(e, 0.99) < 2e-3, (np.quantile(e, 0.5), np.quantile(e, 0.99))
    if mflip.sum() <= 2:
        sc = 0.15 / on["stats"][1]
        assert _rel(g["rendered_normal"].reshape(3, -1) / 3.0, sc * on["g_normal"]) < 5e-3
        assert _rel(g["rendered_distance"].reshape(-1) / 3.0, sc * on["g_dist"]) < 5e-3


def test_out_all_map_argument_gives_the_same_loss_and_gradients():
    """plane_multiview_loss(..., None, None, ..., out_all_map=oam) == the call with the two slices oam[0:3], oam[4:5]; the gradient arrives as one
    (5,H,W) tensor with a zero alpha channel."""
    import mv_cases
    from gsrast.losses import multiview_cfg, plane_multiview_loss
    c = mv_cases.plane_pair(W=96, H=64, seed=3, tex=2.0)
    dev = "cuda:0"
    t = lambda a: torch.tensor(a, device=dev)
    cfg = multiview_cfg(mv_cases.cam_ns(c["view"]), mv_cases.cam_ns(c["near"]), c["W"], c["H"], near_size=(c["W"], c["H"]))
    idx = torch.arange(0, c["W"] * c["H"], 3, dtype=torch.int32, device=dev)
    res = []
    for whole in (False, True):
        pd = t(c["plane_depth"]).requires_grad_(True); pd2 = t(c["near_plane_depth"]).requires_grad_(True)
        oam = torch.cat([t(c["rendered_normal"]), torch.ones(1, c["H"], c["W"], device=dev), t(c["rendered_distance"])], dim=0).requires_grad_(True)
        if whole:
            geo, ncc = plane_multiview_loss(pd, pd2, None, None, t(c["gray"]), t(c["near_gray"]), cfg, indices=idx, out_all_map=oam)
        else:
            geo, ncc = plane_multiview_loss(pd, pd2, oam[0:3], oam[4:5], t(c["gray"]), t(c["near_gray"]), cfg, indices=idx)
        (geo + ncc).backward()
        res.append((geo.item(), ncc.item(), pd.grad.clone(), pd2.grad.clone(), oam.grad.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0]) and abs(res[0][1] - res[1][1]) <= 1e-5 * abs(res[0][1])
    for a, b in zip(res[0][2:], res[1][2:]):       # overlapping patches add their gradients with float atomics: the order differs from run to run
        assert (a - b).norm() <= 1e-5 * b.norm()
    assert res[1][4][3].abs().max() == 0 and res[1][4][0:3].abs().max() > 0


@pytest.mark.parametrize("weights", [(1.0, 1.0, 1.0), (0.5, 2.0, 3.0), (1.0, None, 1.0), (None, 1.0, None), (None, None, 2.0)])
def test_plane_losses_one_node_equals_the_separate_calls(weights):
    """gsrast.losses.plane_losses (single-view normal loss + the two multi-view losses as ONE autograd node, their gradients to plane_depth and
    out_all_map summed by gsr_loss_plane_mv_scale's addends) == plane_geo_loss + plane_multiview_loss evaluated separately: the three values and
    the total gradients, for every combination of used / unused losses (an unused one sends no upstream gradient)."""
    import mv_cases
    from gsrast.losses import multiview_cfg, plane_multiview_loss, plane_geo_loss, plane_losses
    c = mv_cases.plane_pair(W=96, H=64, seed=5, tex=2.0)
    dev = "cuda:0"
    t = lambda a: torch.tensor(a, device=dev)
    cfg = multiview_cfg(mv_cases.cam_ns(c["view"]), mv_cases.cam_ns(c["near"]), c["W"], c["H"], near_size=(c["W"], c["H"]))
    idx = torch.arange(0, c["W"] * c["H"], 3, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(1)
    wmap = torch.rand(c["H"], c["W"], generator=g).to(dev)
    K = torch.tensor([[80.0, 0, c["W"] / 2], [0, 80.0, c["H"] / 2], [0, 0, 1]], device=dev)
    rm = torch.inverse(K.double().t()).float()
    res = []
    for fused in (False, True):
        pd = t(c["plane_depth"]).requires_grad_(True); pd2 = t(c["near_plane_depth"]).requires_grad_(True)
        oam = torch.cat([t(c["rendered_normal"]), torch.full((1, c["H"], c["W"]), 0.9, device=dev), t(c["rendered_distance"])], dim=0).requires_grad_(True)
        if fused:
            nrm, geo, ncc = plane_losses(pd, pd2, oam, t(c["gray"]), t(c["near_gray"]), cfg, rm, wmap, 0.015, 0.03, 0.15, indices=idx)
        else:
            nrm = plane_geo_loss(pd, oam, rm, wmap, 0.015)[0]
            geo, ncc = plane_multiview_loss(pd, pd2, None, None, t(c["gray"]), t(c["near_gray"]), cfg, 0.03, 0.15, indices=idx, out_all_map=oam)
        total = sum(w * v for w, v in zip(weights, (nrm, geo, ncc)) if w is not None)
        total.backward()
        z = lambda x, like: torch.zeros_like(like) if x is None else x.clone()
        res.append((nrm.item(), geo.item(), ncc.item(), z(pd.grad, pd), z(pd2.grad, pd2), z(oam.grad, oam)))
    for a, b in zip(res[0][:3], res[1][:3]):
        assert abs(a - b) <= 1e-6 * max(abs(a), 1e-12)
    for a, b in zip(res[0][3:], res[1][3:]):       # float atomics in the multi-view kernels: the summation order differs from run to run
        assert (a - b).norm() <= 1e-5 * a.norm() + 1e-12, ((a - b).norm().item(), a.norm().item())
    assert (weights[0] is None and weights[2] is None) or res[1][5].abs().max() > 0
