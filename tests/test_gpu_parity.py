"""GPU parity tests (pytest -m gpu): the HIP path, called through the drop-in python packages -> C ABI, against the
CPU oracle on identical inputs.  Bars (BASELINE.json north_star): bit-exact for integer/index work (radii,
tiles_touched, sorted instance list, tile ranges, out_observe); <=1e-4 on rendered RGB/depth/normal maps; <=1e-3
relative on gradients."""
import os
import numpy as np
import pytest
import torch

import oracle
import scenes

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4
GRAD_TOL = 1e-3


def _hiprun():
    import hiprun
    return hiprun


def _lists(st, f, **kw):
    """The HIP library's (filtered) tile-instance list against the oracle's: tests/tile_cull.py.  Returns the view with n_contrib in the oracle's positions."""
    import tile_cull
    return tile_cull.reference_view(st, f, **kw)


def _relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _img_close(a, b, tol=RGB_TOL, frac=1e-4, floor=None):
    """<= tol everywhere, except a vanishing fraction of pixels where a 1-ulp difference in exp() flips one of the
    reference's discrete gates (alpha<1/255, T<1e-4; SURVEY §7 hard part 1).  `floor` = the same output from the FMA-contracted
    build of the oracle: where given, the allowed fraction is max(frac, 1.5 x the oracle's own self-difference)."""
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    scale = max(1.0, float(np.abs(b).max()))
    bad = (d > tol * scale).mean()
    if floor is not None:
        frac = max(frac, 1.5 * float((np.abs(np.asarray(floor, np.float64) - np.asarray(b, np.float64)) > tol * scale).mean()))
    assert bad <= frac, f"{bad:.2e} of pixels differ by more than {tol * scale:.1e} (max {d.max():.3e}; allowed {frac:.2e})"


def _counts_close(a, b):
    """out_observe counts (pixel, splat) pairs that pass the FLOAT gate T > 0.5: integer-valued, but a 1-ulp difference in T
    can move single pairs across the gate.  Equal everywhere except a vanishing number of +-1/2 differences."""
    a = np.asarray(a, np.int64); b = np.asarray(b, np.int64)
    d = np.abs(a - b)
    assert d.max() <= 2 and (d > 0).sum() <= max(2, int(1e-3 * a.size)), (int(d.max()), int((d > 0).sum()))


def _grad_close(a, b, tol=GRAD_TOL, frac=2e-3, floor=None):
    """Gradient parity, three criteria:
      (1) relative L2 error <= tol;
      (2) at most `frac` of the elements off by more than tol * max|ref|;
      (3) PER ELEMENT, |a - b| <= tol * |b| + tol * rms(b) on >= 1 - frac of the elements (north_star: "1e-3 on gradients").
    A handful of outliers is inherent: one-ulp differences (FMA contraction, exp) flip discrete gates / the surfel
    rho3d<=rho2d kink for single (pixel, splat) pairs and re-route that pair's whole gradient.  Building the ORACLE
    itself with -ffp-contract=fast moves dL_drotations by 1e-3*max on 8 of 16000 elements (parity).
    `floor` = the same gradient from the FMA-contracted build of the oracle (BASELINE-size cases): each of the three bars then
    becomes max(nominal, 1.5 x what the oracle differs from itself by; 3 x for the untrimmed L2), plus (4) relative L2 without the
    1e-4 worst elements <= max(tol, floor).  At 300k surfels a single edge-on surfel (ray-splat intersection s = p.xy / p.z with
    p.z ~ 0) moves the relative L2 of dL_dmeans3D by 0.16 between the two oracle builds; measured (profiles/r02_full_size_parity*):
    the HIP path sits at about HALF the oracle's self-difference on every gradient of every surfel case."""
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    rms = np.sqrt((b * b).mean())

    def metrics(x):
        e = np.abs(x - b)
        keep = np.argsort(e)[:e.size - int(np.ceil(1e-4 * e.size))]          # drops the 1e-4 worst-conditioned elements
        return (np.linalg.norm(x - b) / (np.linalg.norm(b) + 1e-30), (e > tol * (np.abs(b).max() + 1e-30)).mean(),
                (e > tol * np.abs(b) + tol * rms).mean(), np.linalg.norm((x - b)[keep]) / (np.linalg.norm(b[keep]) + 1e-30))
    l2, out, per, l2t = metrics(a)
    bars = [tol, frac, frac, tol]
    if floor is not None:
        f = metrics(np.asarray(floor, np.float64).reshape(-1))
        # untrimmed L2 at this size is one or two ill-conditioned splats against each other: 3 x floor; everything else 1.5 x / 1 x
        bars = [max(tol, 3.0 * f[0]), max(frac, 1.5 * f[1]), max(frac, 1.5 * f[2]), max(tol, f[3])]
    assert l2 <= bars[0], f"relative L2 error {l2:.2e} > {bars[0]:.2e}"
    assert out <= bars[1], f"{out:.2e} of elements off by more than {tol}*max (max rel {_relerr(a, b):.2e}; allowed {bars[1]:.2e})"
    assert per <= bars[2], f"{per:.2e} of elements fail |a-b| <= {tol}|b| + {tol} rms(b) (allowed {bars[2]:.2e})"
    if floor is not None:
        assert l2t <= bars[3], f"relative L2 error without the 1e-4 worst elements {l2t:.2e} > {bars[3]:.2e}"


def _ncontrib_close(nc_hip, nc_ref, ft_hip, ft_ref):
    """n_contrib (index of the last contributing splat) is bit-exact except where a float gate sits within rounding of its threshold:
    (A) the termination gate test_T < 1e-4: one side accepted a splat the other stopped at, so the smaller final T lies within
        1e-6 of 1e-4; or
    (B) a gate on the LAST contributor itself (alpha < 1/255, the surfel depth / p.z gates): one side blends one more splat, and the
        two final T differ by that splat's (1 - alpha) -- both sides then agree on every earlier splat, so T_a / T_b is within
        [0.01, 1) (alpha <= 0.99) but NOT equal.
    Anything else (equal T but different index, ...) is a logic divergence and fails.  Returns the mismatch fraction."""
    nc_hip = np.asarray(nc_hip); nc_ref = np.asarray(nc_ref)
    bad = nc_hip != nc_ref
    if bad.any():
        a = np.asarray(ft_hip, np.float64)[bad]; b = np.asarray(ft_ref, np.float64)[bad]
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        gate_a = np.abs(lo - 1e-4) <= 1e-6
        gate_b = (lo < hi) & (lo >= 0.0099 * hi)
        unexplained = ~(gate_a | gate_b)
        assert not unexplained.any(), f"{int(unexplained.sum())} n_contrib mismatches away from every float gate"
    frac = float(bad.mean())
    assert frac <= 1e-4, f"n_contrib differs on {frac:.2e} of the pixels"
    return frac


CASES = [
    ("ewa", "precomp", 4000, 256, 160, 0),
    ("ewa", "sh", 6000, 330, 190, 1),
    ("plane", "precomp", 4000, 256, 160, 0),
    ("plane", "sh", 3000, 200, 120, 1),
    ("surfel", "precomp", 4000, 256, 160, 0),
    ("surfel", "sh", 6000, 330, 190, 1),
]


@pytest.mark.parametrize("variant,cm,P,W,H,pose", CASES)
def test_forward_backward_parity(variant, cm, P, W, H, pose):
    hr = _hiprun()
    sc = scenes.make_scene(variant, P, W, H, seed=11, color_mode=cm, bg=(0.2, 0.4, 0.6), pose=pose)
    og = scenes.random_out_grads(variant, W, H, seed=11, scale=1.0)
    with oracle.Forward(sc, variant) as f:
        g = f.backward(**og)
        st = hr.run_raw(variant, sc)
        # ---- integer stages: bit-exact
        assert np.array_equal(st["radii"], f.radii)
        view = _lists(st, f, variant=variant)      # the list is the oracle's minus instances that reach no pixel; tiles_touched, ranges follow it
        if os.environ.get("GSR_TILE_CULL") == "0":
            assert st["R"] == f.R
            assert np.array_equal(st["tiles_touched"], f.tiles_touched())
            assert np.array_equal(st["point_list"], f.point_list())
            assert np.array_equal(st["tile_keys"], (f.keys() >> np.uint64(32)).astype(np.uint32))
            rr = f.ranges(); touched = rr[:, 1] > rr[:, 0]
            assert np.array_equal(st["ranges"][touched], rr[touched])
        assert np.all(np.maximum(st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0].astype(np.int64), 0)[f.ranges()[:, 1] <= f.ranges()[:, 0]] == 0)
        ft, nc = f.image_state()
        _ncontrib_close(view["n_contrib"][0], nc[0], st["final_T"][0], ft[0])
        if variant == "surfel":
            assert (view["n_contrib"][1] == nc[1]).mean() > 0.9999          # median contributor: behind the T > 0.5 gate
        # ---- images
        _img_close(st["color"], f.color)
        _img_close(st["final_T"], ft)
        if variant == "surfel":
            for ch in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10):
                _img_close(st["others"][ch], f.others[ch])
            assert (st["others"][7] == f.others[7]).mean() > 0.9999         # median splat index
        if variant == "plane":
            _counts_close(st["observe"], f.observe)
            _img_close(st["all_map"], f.out_all_map)
            _img_close(st["plane_depth"], f.plane_depth, frac=1e-3)
        # ---- gradients through the public autograd API
        res = hr.run(variant, sc, og)
    gg = res["grads"]
    pairs = [("dL_dmeans3D", "dL_dmeans3D"), ("dL_dscales", "dL_dscales"), ("dL_drotations", "dL_drotations"),
             ("dL_dopacities", "dL_dopacity"), ("dL_dmeans2D", "dL_dmeans2D")]
    pairs.append(("dL_dshs", "dL_dsh") if cm == "sh" else ("dL_dcolors_precomp", "dL_dcolors"))
    if variant == "plane":
        pairs += [("dL_dall_map", "dL_dall_map"), ("dL_dmeans2D_abs", "dL_dmeans2D_abs")]
    for a, b in pairs:
        _grad_close(gg[a], g[b])


# (active degree, coefficients per gaussian).  M = 16 with degree 0 / 1 / 2 is the vanilla model at iterations 1 ... 3000 (base_gaussian.py:45,
# vanilla_gaussian.py:440-442: every model allocates (max_sh_degree + 1)^2 = 16 coefficients and raises active_sh_degree every 1000 steps) and goes
# through the LDS-staged SH16 kernels; M = 1 / 4 / 9 (a model built with max_sh_degree 0 / 1 / 2; M = shs.size(1), 3DGS/rasterize_points.cu:66-70)
# takes the generic kernels (k_preprocess_* <false>, k_preprocess_bwd_* <false>), also with the degree below what M allows.
SH_CASES = [(0, 16), (1, 16), (2, 16), (0, 1), (1, 4), (2, 9), (0, 4), (1, 9)]


@pytest.mark.parametrize("deg,M", SH_CASES)
@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_sh_degrees_and_coefficient_counts(variant, deg, M):
    """3DGS/forward.cu:20-71 (computeColorFromSH: bands up to the ACTIVE degree, +0.5, clamp at 0 with a per-channel mask), backward.cu:20-139 (dL_dsh
    for the active bands only -- the rows above stay at the zeros the glue allocated --, the clamp mask zeroing a channel's gradient, and the
    view-direction term added to dL_dmeans3D from degree 1 on).  SURFEL / PLANE carry copies of both functions."""
    hr = _hiprun()
    P, W, H = 3000, 208, 128
    sc = scenes.make_scene(variant, P, W, H, seed=20 + deg, color_mode="sh", sh_degree=deg, sh_M=M, bg=(0.1, 0.0, 0.3), pose=deg % 2)
    assert sc["shs"].shape == (P, M, 3) and sc["sh_degree"] == deg
    sc["shs"][:, 1:] *= 8.0                                           # strong view dependence: the direction term of dL_dmeans3D is 1e-2-class, not 1e-4
    og = scenes.random_out_grads(variant, W, H, seed=20 + deg, scale=1.0)
    # every output and gradient (dL_dsh included) against the float64 truth; the integer stages and the instance list against the float32 oracle
    rep = _check_against_truth(hr, variant, "sh", sc, og)
    with oracle.Forward(sc, variant) as f:
        g = f.backward(**og)
        vis = f.radii > 0
    gg = rep["hip_grads"]
    used = (deg + 1) ** 2
    dsh = gg["dL_dshs"]
    assert dsh.shape == (P, M, 3)
    assert not dsh[:, used:].any()                                    # exact zeros above the active degree (and nothing written past them)
    assert not dsh[~vis].any()                                        # culled gaussians: the rows the reference never touches
    assert np.abs(dsh[vis][:, :used]).max() > 0
    # clamp mask: a channel whose colour was clamped at 0 receives no gradient in ANY band -- same (gaussian, channel) set as the oracle's
    dead_hip = ~dsh[vis][:, :used].any(axis=1)
    dead_ref = ~g["dL_dsh"][vis][:, :used].any(axis=1)
    assert dead_ref.any() and not dead_ref.all()                      # the scene exercises both sides of the clamp
    assert (dead_hip != dead_ref).mean() < 2e-3                       # (a channel can also be dead because no pixel sent it a gradient: same on both sides up to gate flips)
    if deg > 0:
        # the view-direction term exists from degree 1 on: the same gaussians with their (clamped) colours handed over as colors_precomp get
        # the same image and a DIFFERENT dL_dmeans3D; the difference is the term, and it must match the oracle's difference
        with oracle.Forward(sc, variant) as f:
            rgb = f.geom()["rgb"].copy()
        sc2 = {k: v for k, v in sc.items() if k != "shs"}
        sc2["colors_precomp"] = rgb; sc2["sh_degree"] = 0
        with oracle.Forward(sc2, variant) as f2:
            g2 = f2.backward(**og)
        res2 = hr.run(variant, sc2, og)
        _img_close(res2["color"], rep["hip_color"], tol=2e-6)
        term_hip = gg["dL_dmeans3D"].astype(np.float64) - res2["grads"]["dL_dmeans3D"]
        term_ref = g["dL_dmeans3D"].astype(np.float64) - g2["dL_dmeans3D"]
        gn = np.linalg.norm(g["dL_dmeans3D"])
        assert np.linalg.norm(term_ref) > 1e-3 * gn
        # both runs of a side share the blend, so the difference IS the term (+ the summation-order noise of two float32 gradients, 1e-5-class of |g|)
        err = np.linalg.norm(term_hip - term_ref)
        assert err <= 2 * GRAD_TOL * np.linalg.norm(term_ref) + 2e-5 * gn, (err, np.linalg.norm(term_ref), gn)


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_sh16_coefficients_at_an_unaligned_address_take_the_generic_kernels(variant):
    """The SH16 kernels need 16-byte aligned rows (gsr_preprocess.hip: sh16 = M == 16 && aligned); a (P,16,3) VIEW that starts 4 bytes into its
    storage must give bit-identical colours and the same dL_dsh through the generic kernels."""
    hr = _hiprun()
    P, W, H = 2000, 160, 112
    sc = scenes.make_scene(variant, P, W, H, seed=31, color_mode="sh", sh_degree=3)
    og = scenes.random_out_grads(variant, W, H, seed=31, scale=1.0)
    t = hr.to_dev(sc)
    rs = hr.settings(variant, t)
    gcol = torch.from_numpy(og["dL_dcolor"]).cuda()
    mod = {"ewa": hr.dgr, "surfel": hr.dsr, "plane": hr.dpr}[variant]

    def render(shs):
        kw = dict(means3D=t["means3D"], means2D=torch.zeros((P, 3), device="cuda"), opacities=t["opacities"], shs=shs, colors_precomp=None,
                  scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        if variant == "plane":
            kw.update(means2D_abs=torch.zeros((P, 3), device="cuda"), all_map=t["all_map"])
        out = mod.GaussianRasterizer(rs)(**kw)
        (out[0] * gcol).sum().backward()
        return out[0].detach().clone(), out[1].clone()

    leaf = t["shs"].clone().requires_grad_(True)
    c0, r0 = render(leaf)
    store = torch.zeros(P * 48 + 1, dtype=torch.float32, device="cuda")
    store[1:] = t["shs"].reshape(-1)
    store.requires_grad_(True)
    un = store[1:].view(P, 16, 3)
    assert un.data_ptr() % 16 == 4 and un.is_contiguous()
    c1, r1 = render(un)
    assert torch.equal(c0, c1) and torch.equal(r0, r1)
    # dL_dsh is a function of dL_dcolors, which the blend backward sums with float atomics: equal up to the summation order of two launches
    ga, gb = store.grad[1:].view(P, 16, 3), leaf.grad
    assert store.grad[0] == 0 and torch.equal(ga == 0, gb == 0)
    assert torch.linalg.norm(ga - gb) <= 1e-5 * torch.linalg.norm(gb)


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
@pytest.mark.parametrize("seed,W,H,fx,fy,sigma,bg", [
    (101, 400, 225, 333.0, 333.0, 4.0, (0.0, 0.0, 0.0)),      # the loader's 1600x900 cap, scaled
    (102, 97, 61, 70.0, 95.0, 2.5, (1.0, 1.0, 1.0)),          # odd size, fx != fy, white background
    (103, 512, 48, 300.0, 300.0, 7.0, (0.3, 0.0, 0.9)),       # wide strip: many tile columns, 3 rows
])
def test_varied_cameras(variant, seed, W, H, fx, fy, sigma, bg):
    hr = _hiprun()
    sc = scenes.make_scene(variant, 2500, W, H, fx=fx, fy=fy, seed=seed, sigma_px=sigma, bg=bg, pose=seed % 2)
    og = scenes.random_out_grads(variant, W, H, seed=seed, scale=1.0)
    with oracle.Forward(sc, variant) as f:
        g = f.backward(**og)
        st = hr.run_raw(variant, sc)
        assert np.array_equal(st["radii"], f.radii)
        _lists(st, f, variant=variant)
        _img_close(st["color"], f.color)
        if variant == "surfel":
            for ch in (0, 1, 2, 3, 4, 5, 6):
                _img_close(st["others"][ch], f.others[ch])
        if variant == "plane":
            _counts_close(st["observe"], f.observe)
            _img_close(st["plane_depth"], f.plane_depth, frac=1e-3)
        res = hr.run(variant, sc, og)
    for a, b in (("dL_dmeans3D", "dL_dmeans3D"), ("dL_dscales", "dL_dscales"), ("dL_drotations", "dL_drotations"),
                 ("dL_dopacities", "dL_dopacity"), ("dL_dcolors_precomp", "dL_dcolors")):
        _grad_close(res["grads"][a], g[b])


def test_precomputed_cov3d_and_transmat():
    hr = _hiprun()
    # EWA with cov3D_precomp
    sc = scenes.make_scene("ewa", 2000, 160, 112, seed=3)
    with oracle.Forward(sc, "ewa") as f:
        cov = f.geom()["cov"].copy()
    sc2 = {k: v for k, v in sc.items() if k not in ("scales", "rotations")}
    sc2["cov3D_precomp"] = cov
    og = scenes.random_out_grads("ewa", 160, 112, seed=3, scale=1.0)
    with oracle.Forward(sc2, "ewa") as f:
        g = f.backward(**og)
        res = hr.run("ewa", sc2, og)
        _img_close(res["color"], f.color)
        _grad_close(res["grads"]["dL_dcov3D_precomp"], g["dL_dcov3D"])
        _grad_close(res["grads"]["dL_dmeans3D"], g["dL_dmeans3D"])
    # SURFEL with transMat_precomp (cov3D_precomp slot carries (P,9))
    ss = scenes.make_scene("surfel", 2000, 160, 112, seed=4)
    with oracle.Forward(ss, "surfel") as f:
        T = f.geom()["cov"].copy()
        vis = f.radii > 0
    ss2 = {k: v for k, v in ss.items() if k not in ("scales", "rotations")}
    ss2["cov3D_precomp"] = T
    ss2["means3D"] = ss["means3D"][vis]; ss2["opacities"] = ss["opacities"][vis]
    ss2["colors_precomp"] = ss["colors_precomp"][vis]; ss2["cov3D_precomp"] = T[vis]
    og = scenes.random_out_grads("surfel", 160, 112, seed=4, scale=1.0)
    with oracle.Forward(ss2, "surfel") as f:
        g = f.backward(**og)
        res = hr.run("surfel", ss2, og)
        _img_close(res["color"], f.color)
        assert np.array_equal(res["radii"], f.radii)
        _grad_close(res["grads"]["dL_dcov3D_precomp"], g["dL_dcov3D"])


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_edge_cases(variant):
    hr = _hiprun()
    # ragged image (not a multiple of 16 or 8), huge and tiny splats, scale_modifier, nothing visible, P == 0
    sc = scenes.make_scene(variant, 1500, 131, 77, seed=5, sigma_px=9.0, scale_modifier=1.3, bg=(1.0, 0.0, 0.5))
    sc["opacities"][:50] = 0.999      # exercises the 0.99 clamp
    sc["opacities"][50:100] = 0.002   # below 1/255: never contributes
    og = scenes.random_out_grads(variant, 131, 77, seed=5, scale=1.0)
    if variant == "surfel":
        og["dL_dothers"][8:11] = 0.25   # median-normal quirk path
    with oracle.Forward(sc, variant) as f:
        g = f.backward(**og)
        res = hr.run(variant, sc, og)
        assert np.array_equal(res["radii"], f.radii)
        _img_close(res["color"], f.color)
        _grad_close(res["grads"]["dL_dmeans3D"], g["dL_dmeans3D"])
        _grad_close(res["grads"]["dL_dscales"], g["dL_dscales"])
        _grad_close(res["grads"]["dL_dopacities"], g["dL_dopacity"])
    # everything behind the camera -> R == 0, background only
    sb = dict(sc); sb["means3D"] = sc["means3D"].copy(); sb["means3D"][:, 2] = -5.0
    res = hr.run(variant, sb, og)
    assert (res["radii"] == 0).all()
    assert np.allclose(res["color"], np.asarray(sc["bg"])[:, None, None], atol=1e-7)
    assert all(np.abs(v).max() == 0 for v in res["grads"].values() if v is not None)
    # P == 0 (reference returns zero-initialised outputs, rasterize_points.cu:79)
    s0 = {k: (v[:0] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 1500 else v) for k, v in sc.items()}
    res = hr.run(variant, s0)
    assert res["color"].shape == (3, 77, 131) and np.abs(res["color"]).max() == 0


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_speculative_forward_matches_exact_and_survives_overflow(variant):
    """gsr_forward (stage 2 enqueued against a capacity guessed from the previous call) must give bit-identical results
    to stage1+stage2, including when the guess is too small (overflow -> redo; PLANE re-zeroes out_observe)."""
    from gsrast import rasterize as rz
    hr = _hiprun()
    W, H = 208, 144
    small = scenes.make_scene(variant, 300, W, H, seed=21, sigma_px=2.0)
    big = scenes.make_scene(variant, 9000, W, H, seed=22, sigma_px=12.0)      # R(big) >> 1.25 * R(small) + 16384
    key = (torch.cuda.current_device(), hr.VID[variant], W, H)
    rz._R_HINT.pop(key, None)
    exact = hr.run_raw(variant, big)                                           # no hint -> stage1 + stage2
    assert rz._R_HINT[key] == exact["R"] and exact["R"] > 60000
    spec = hr.run_raw(variant, big)                                            # hint fits -> single call
    rz._R_HINT[key] = 100                                                      # force an overflow on the next call
    ovf = hr.run_raw(variant, big)
    for other in (spec, ovf):
        assert other["R"] == exact["R"]
        for k in ("color", "radii", "point_list", "tile_keys", "ranges", "final_T", "n_contrib"):
            assert np.array_equal(other[k], exact[k]), k
        if variant == "plane":
            assert np.array_equal(other["observe"], exact["observe"])
    hr.run_raw(variant, small)                                                 # shrinking is fine too
    with oracle.Forward(big, variant) as f:
        _lists(ovf, f)
        _img_close(ovf["color"], f.color)


def test_render_geo_false_plane():
    hr = _hiprun()
    sc = scenes.make_scene("plane", 1500, 128, 96, seed=6); sc["render_geo"] = False
    og = dict(dL_dcolor=scenes.random_out_grads("plane", 128, 96, seed=6, scale=1.0)["dL_dcolor"])
    with oracle.Forward(sc, "plane") as f:
        g = f.backward(**og)
        res = hr.run("plane", sc, og)
        _img_close(res["color"], f.color)
        _counts_close(res["observe"], f.observe)                    # counted regardless of render_geo
        assert np.abs(res["out_all_map"]).max() == 0 and np.abs(res["plane_depth"]).max() == 0
        _grad_close(res["grads"]["dL_dmeans3D"], g["dL_dmeans3D"])


def test_visible_filter_mark_visible_dist2_tsdf():
    import scaffold_filter
    from simple_knn._C import distCUDA2
    from gsrast.tsdf import tsdf_integrate_
    hr = _hiprun()
    sc = scenes.make_scene("ewa", 20000, 640, 360, seed=7)
    t = hr.to_dev(sc)
    rs = scaffold_filter.GaussianRasterizationSettings(
        image_height=360, image_width=640, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=t["bg"], scale_modifier=1.0,
        viewmatrix=t["viewmatrix"], projmatrix=t["projmatrix"], sh_degree=0, campos=t["campos"], prefiltered=False, debug=False)
    radii = scaffold_filter.GaussianRasterizer(rs).visible_filter(t["means3D"], scales=t["scales"], rotations=t["rotations"])
    assert np.array_equal(radii.cpu().numpy(), oracle.visible_filter(sc))
    import diff_gaussian_rasterization as dgr
    mv = dgr.GaussianRasterizer(hr.settings("ewa", t)).markVisible(t["means3D"])
    assert np.array_equal(mv.cpu().numpy(), oracle.mark_visible(sc["means3D"], sc["viewmatrix"], sc["projmatrix"]))
    # distCUDA2
    pts = sc["means3D"][:3000]
    d = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    assert np.allclose(d, oracle.dist2(pts), rtol=1e-5, atol=0)
    # TSDF: three frames into the same grid
    rng = np.random.default_rng(0)
    W, H = 96, 64
    V = 40003        # not a multiple of 4: exercises the vector path and the scalar tail
    grid = rng.uniform(-1, 1, (V, 3)).astype(np.float32) * np.array([2.0, 1.2, 1.0], np.float32) + np.array([0, 0, 5.0], np.float32)
    tsdf = np.ones(V, np.float32); wgt = np.ones(V, np.float32); rgb = np.zeros((V, 3), np.float32)
    tg = torch.ones(V, device="cuda"); wg = torch.ones(V, device="cuda"); cg = torch.zeros((V, 3), device="cuda")
    for fr in range(3):
        cam = scenes.make_camera(W, H, 80.0, 80.0, yaw_deg=5.0 * fr, t=(0.1 * fr, 0, 0))
        depth = rng.uniform(4.0, 6.0, (1, H, W)).astype(np.float32)
        col = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
        oracle.tsdf_integrate(grid, cam["projmatrix"], depth, col, 0.3, tsdf, wgt, rgb)
        tsdf_integrate_(torch.from_numpy(grid).cuda(), torch.from_numpy(cam["projmatrix"]).cuda(), torch.from_numpy(depth).cuda(),
                        torch.from_numpy(col).cuda(), 0.3, tg, cg, wg)
    assert np.array_equal(wg.cpu().numpy(), wgt)
    assert np.abs(tg.cpu().numpy() - tsdf).max() < 1e-4      # float32 bilinear interpolation with cancellation (d - z)
    assert np.abs(cg.cpu().numpy() - rgb).max() < 1e-4
    assert (wgt > 1).sum() > 1000


def test_fused_loss_matches_torch():
    from gsrast.losses import l1_plus_linear
    g = torch.Generator(device="cpu").manual_seed(0)
    for shape_c, shape_a in (((3, 77, 131), (11, 77, 131)), ((3, 64, 64), None)):
        c = torch.rand(shape_c, generator=g).cuda().requires_grad_(True)
        gt = torch.rand(shape_c, generator=g).cuda()
        a = torch.randn(shape_a, generator=g).cuda().requires_grad_(True) if shape_a else None
        w = torch.randn(shape_a, generator=g).cuda() if shape_a else None
        ref = (c - gt).abs().mean() + ((a * w).sum() if a is not None else 0.0)
        gr = torch.autograd.grad(ref * 1.7, [c] + ([a] if a is not None else []))
        out = l1_plus_linear(c, gt, a, w)
        go = torch.autograd.grad(out * 1.7, [c] + ([a] if a is not None else []))
        assert abs(float(out.detach()) - float(ref.detach())) <= 1e-4 * max(1.0, abs(float(ref.detach())))
        for x, y in zip(go, gr):
            assert torch.allclose(x, y, atol=1e-7, rtol=1e-5)


def test_dist2_morton_pruned_is_exact_and_dense_tsdf():
    from simple_knn._C import distCUDA2
    from gsrast.tsdf import DenseTSDFVolume
    rng = np.random.default_rng(3)
    # clustered + duplicated points: the box pruning must still return the exact 3-NN
    pts = np.concatenate([rng.normal(0, 1, (20000, 3)), rng.normal(5, 0.01, (5000, 3)), np.zeros((7, 3))]).astype(np.float32)
    d = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    assert np.array_equal(d, oracle.dist2(pts))
    # dense Open3D-style volume, 4 frames
    dims, origin, vl, trunc = (48, 40, 56), (-1.2, -1.0, 3.0), 0.05, 0.25
    vol = DenseTSDFVolume(origin, vl, dims, trunc)
    t = np.zeros(dims, np.float32); w = np.zeros(dims, np.float32); c = np.zeros(dims + (3,), np.float32)
    W, H, fx, fy = 128, 96, 110.0, 105.0
    for fr in range(4):
        a = 0.1 * fr
        E = np.array([[np.cos(a), 0, np.sin(a), 0.05 * fr], [0, 1, 0, -0.02 * fr], [-np.sin(a), 0, np.cos(a), 0.1], [0, 0, 0, 1]], np.float32)
        depth = rng.uniform(3.5, 5.5, (1, H, W)).astype(np.float32)
        depth[0, :5] = 0.0
        rgb = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
        q = (np.clip(rgb, 0, 1) * 255).astype(np.uint8).astype(np.float32)
        oracle.tsdf_integrate_dense(dims, origin, vl, trunc, 5.2, depth, q, fx, fy, W / 2, H / 2, E, t, w, c)
        vol.integrate(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), fx, fy, W / 2, H / 2, E, depth_trunc=5.2)
    assert np.array_equal(vol.weight.cpu().numpy(), w) and (w > 0).sum() > 5000
    assert np.abs(vol.tsdf.cpu().numpy() - t).max() < 1e-4        # v_rcp / v_sqrt in the dense kernel
    assert np.abs(vol.color.cpu().numpy() - c).max() < 1e-3


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_full_size_properties(variant):
    """BASELINE full size (300k gaussians, 1920x1080): size-independent properties instead of the (slow) oracle."""
    hr = _hiprun()
    P, W, H = 300000, 1920, 1080
    sc = scenes.make_scene(variant, P, W, H, seed=0)
    st = hr.run_raw(variant, sc)
    R = st["R"]
    assert R == int(st["tiles_touched"].sum())
    tk, pl = st["tile_keys"].astype(np.int64), st["point_list"].astype(np.int64)
    assert np.all(np.diff(tk) >= 0)                                           # sorted by tile
    pv = sc["means3D"] @ sc["viewmatrix"][:3, :3] + sc["viewmatrix"][3, :3]
    depth_bits = pv[:, 2].astype(np.float32).view(np.uint32).astype(np.int64)
    same = np.nonzero(np.diff(tk) == 0)[0]
    d0, d1 = depth_bits[pl[same]], depth_bits[pl[same + 1]]
    assert np.all(d0 <= d1)                                                   # then by depth
    assert np.all(pl[same][d0 == d1] < pl[same + 1][d0 == d1])                # ties by gaussian id (stable)
    counts = np.bincount(tk, minlength=st["ranges"].shape[0])
    assert np.array_equal(st["ranges"][:, 1] - st["ranges"][:, 0], counts)    # ranges == histogram
    nz = counts > 0
    assert np.array_equal(st["ranges"][nz, 0], (np.cumsum(counts) - counts)[nz])
    assert np.array_equal(np.bincount(pl, minlength=P), st["tiles_touched"])  # every instance emitted exactly once
    assert (st["final_T"][0] >= 0).all() and (st["final_T"][0] <= 1).all()
    assert (st["n_contrib"][0] <= counts[(np.arange(H)[:, None] // 16) * ((W + 15) // 16) + np.arange(W)[None] // 16]).all()
    if variant == "surfel":
        assert np.abs(st["others"][1] - (1 - st["final_T"][0])).max() < 1e-6
    # linearity in colour + adjoint identity <R(c), g> == <c, R^T g> (bg = 0): ties forward and backward together
    og = scenes.random_out_grads(variant, W, H, seed=0, scale=1.0)
    only_color = dict(dL_dcolor=og["dL_dcolor"])
    res = hr.run(variant, sc, only_color)
    lhs = float((res["color"].astype(np.float64) * og["dL_dcolor"]).sum())
    rhs = float((sc["colors_precomp"].astype(np.float64) * res["grads"]["dL_dcolors_precomp"]).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    sc2 = dict(sc); sc2["colors_precomp"] = (2.0 * sc["colors_precomp"]).astype(np.float32)
    res2 = hr.run(variant, sc2)
    assert np.abs(res2["color"] - 2.0 * res["color"]).max() < 2e-5
    # determinism of the forward
    res3 = hr.run(variant, sc)
    assert np.array_equal(res3["color"], res["color"])


# ---- BASELINE full size against the oracle: 300k gaussians, 1920x1080, seeds {0,1,2}, both poses, both colour modes.
# The oracle runs the whole workload in ~2 s on the GPU host's cores, so there is no reason to stop at properties.
FULL_CASES = [
    ("surfel", "precomp", 0, 0), ("ewa", "precomp", 0, 0), ("plane", "precomp", 0, 0),
    # "sh:d" = (P,16,3) coefficients with active degree d < 3: the vanilla model before iteration 3000 (vanilla_gaussian.py:440-442)
    ("ewa", "sh:0", 3, 0), ("surfel", "sh:1", 3, 1), ("plane", "sh:2", 3, 1),
]
# the other seeds / poses / degree-3 SH cases of rounds 2-4 run in tools/full_parity_report.py (profiles/r05_full_size_parity.jsonl holds all 14): every
# case costs 10-20 s of oracle time (float32, float64 truth, float64 floor, the brute-force alpha maximum of every dropped instance) on the GPU box's host
FULL_CASES_REPORT = FULL_CASES + [("surfel", "precomp", 1, 0), ("surfel", "precomp", 2, 1), ("surfel", "sh", 1, 1), ("ewa", "sh", 1, 0), ("ewa", "precomp", 2, 1),
                                  ("ewa", "sh", 0, 1), ("plane", "precomp", 1, 1), ("plane", "sh", 2, 0)]


def _hip_outputs(hr, variant, sc, og):
    """The HIP library's results in the layout tests/parity_truth.py compares (integer stages included)."""
    st = hr.run_raw(variant, sc)
    res = hr.run(variant, sc, og)
    cand = dict(color=st["color"], final_T=st["final_T"], n_contrib=st["n_contrib"], grads=res["grads"])
    if variant == "surfel":
        cand["others"] = st["others"]
    if variant == "plane":
        cand.update(all_map=st["all_map"], plane_depth=st["plane_depth"], observe=st["observe"])
    return st, cand


def _check_against_truth(hr, variant, cm, sc, og):
    """Integer stages bit-exact against the float32 oracle; everything else against the FLOAT64 truth (tests/parity_truth.py): exact indices and
    the nominal 1e-4 on every pixel whose gate decisions are robust under float32 rounding, every other mismatch attributed to a named gate,
    gradients within max(1e-3, 2 x the float32 oracle's own error vs the truth)."""
    import parity_truth as pt
    st, cand = _hip_outputs(hr, variant, sc, og)
    f32, fma, truth, ints = pt.run_oracles(sc, variant, og, hip_state=st, fma=sc["means3D"].shape[0] <= 50000)     # incl. the filtered instance list against the oracle's (tests/tile_cull.py)
    assert np.array_equal(st["radii"], ints["radii"])
    cand["n_contrib"] = ints["view"]["n_contrib"]                             # positions in the oracle's list
    rep = pt.check_case(variant, cm, cand, f32, fma, truth)
    rep["tile_instances"] = {k: v for k, v in ints["view"].items() if k not in ("keep", "n_contrib")}
    rep["hip_grads"] = cand["grads"]; rep["hip_color"] = cand["color"]
    return rep


@pytest.mark.parametrize("variant,cm,seed,pose", FULL_CASES)
def test_full_size_oracle_parity(variant, cm, seed, pose):
    """BASELINE size (P = 300 000, 1920x1080).  Round 3: the bars are no longer relaxed to the oracle's FMA self-difference -- the candidate is
    compared with a float64 evaluation of the same computation, see _check_against_truth."""
    hr = _hiprun()
    P, W, H = 300000, 1920, 1080
    cm, _, deg = cm.partition(":")
    sc = scenes.make_scene(variant, P, W, H, seed=seed, color_mode=cm, pose=pose, bg=(0.1, 0.3, 0.2) if seed else (0.0, 0.0, 0.0),
                           sh_degree=int(deg or 3))
    og = scenes.random_out_grads(variant, W, H, seed=seed)           # N(0,1)/N
    rep = _check_against_truth(hr, variant, cm, sc, og)
    assert rep["robust_pixel_fraction"] > 0.95
    if deg:
        assert not rep["hip_grads"]["dL_dshs"][:, (int(deg) + 1) ** 2:].any()


@pytest.mark.parametrize("variant,cm,seed,pose", FULL_CASES_REPORT[len(FULL_CASES):])
def test_full_size_oracle_parity_remaining_report_cases(variant, cm, seed, pose):
    """The other eight BASELINE-size cases (seeds 1 / 2, both poses, degree-3 SH) with the suite's own assertion: 15-28 s of oracle time each, so they run when
    GSR_FULL_PARITY=1 is set -- and, in any case, in tools/full_parity_report.py, which exits non-zero on a violated bar and is step 6 of tools/gpu_profile_r06.sh
    (profiles/r06_full_size_parity.jsonl holds all 17 cases of this round)."""
    if os.environ.get("GSR_FULL_PARITY") != "1":
        pytest.skip("set GSR_FULL_PARITY=1 (or run tools/full_parity_report.py) for the remaining BASELINE-size cases")
    test_full_size_oracle_parity(variant, cm, seed, pose)


# BASELINE size on a NON-UNIFORM scene: scenes.concentrate pulls a fraction of the gaussians towards the optical axis, so that a few hundred tiles carry
# lists of 1 500 ... 12 000 entries.  The paths only long lists take -- chunk halving and longest-first launch order in the splat-parallel backward
# (csrc/gsr_blend_sp.hip), k_tile_order / the global-order feedback, the bitonic and radix fallbacks of the per-tile depth sort (csrc/gsr_tile_sort.h) -- meet the
# float64 truth and the float32-geometry floor here, with the nominal bars of tests/parity_truth.py, not only the integer-list checks.
# (surfel/backward.cu:143-447, rasterizer_impl.cu:300-308 are what the long lists restate.)
SKEW_CASES = [("surfel", 0.6, 0.12), ("ewa", 0.7, 0.05)]


@pytest.mark.parametrize("variant,frac,scale", SKEW_CASES)
def test_full_size_oracle_parity_on_long_tile_lists(variant, frac, scale):
    hr = _hiprun()
    P, W, H = 300000, 1920, 1080
    sc = scenes.concentrate(scenes.make_scene(variant, P, W, H, seed=0), frac, scale)
    og = scenes.random_out_grads(variant, W, H, seed=0)
    rep = _check_against_truth(hr, variant, "precomp", sc, og)
    ln = np.diff(hr.run_raw(variant, sc)["ranges"].astype(np.int64), axis=1)[:, 0]
    assert ln.max() >= 1500 and (ln > 1024).sum() >= 50, (int(ln.max()), int((ln > 1024).sum()))      # the long-list paths were taken
    assert rep["robust_pixel_fraction"] > 0.9


@pytest.mark.parametrize("variant,cm,P,W,H,pose", CASES)
def test_small_cases_against_the_float64_truth(variant, cm, P, W, H, pose):
    hr = _hiprun()
    sc = scenes.make_scene(variant, P, W, H, seed=11, color_mode=cm, bg=(0.2, 0.4, 0.6), pose=pose)
    og = scenes.random_out_grads(variant, W, H, seed=11, scale=1.0)
    _check_against_truth(hr, variant, cm, sc, og)


@pytest.mark.parametrize("variant", ["surfel", "plane"])
def test_unused_outputs_send_no_gradient_tensor(variant):
    """The autograd bridges do not ask autograd to materialise zero gradients for outputs the loss never touched (set_materialize_grads(False)):
    such an output reaches the kernels as a null pointer.  The gradients must equal those of the same loss written with explicit zero
    weights on every output (what the reference's materialised zeros amount to)."""
    hr = _hiprun()
    import torch
    W, H = 200, 120
    sc = scenes.make_scene(variant, 2500, W, H, seed=77)
    t = hr.to_dev(sc, "cuda")
    rs = hr.settings(variant, t)
    gsel = torch.Generator().manual_seed(5)
    wsel = torch.rand((1, H, W), generator=gsel).cuda()

    def run(explicit_zeros):
        leaves = {k: t[k].clone().requires_grad_(True) for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations")}
        m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
        kw = dict(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"], colors_precomp=leaves["colors_precomp"],
                  scales=leaves["scales"], rotations=leaves["rotations"])
        if variant == "surfel":
            color, radii, others = hr.dsr.GaussianRasterizer(rs)(**kw)
            loss = (others[0:1] * wsel).sum()                      # depth channel only: `color` is never used
            if explicit_zeros:
                loss = loss + (color * 0.0).sum()
        else:
            m2a = torch.zeros_like(leaves["means3D"], requires_grad=True)
            am = t["all_map"].clone().requires_grad_(True); leaves["all_map"] = am
            color, radii, obs, oam, pd = hr.dpr.GaussianRasterizer(rs)(means2D_abs=m2a, all_map=am, **kw)
            loss = (pd * wsel).sum()                               # plane depth only: colour and all_map maps unused
            if explicit_zeros:
                loss = loss + (color * 0.0).sum() + (oam * 0.0).sum()
        loss.backward()
        return {k: v.grad.detach().cpu().numpy() for k, v in leaves.items()}, m2.grad.detach().cpu().numpy()

    ga, ma = run(False)
    gb, mb = run(True)
    for k in ga:
        d = np.linalg.norm(ga[k].astype(np.float64) - gb[k])               # two runs of the backward differ by the order of the float atomics
        assert d <= 1e-4 * np.linalg.norm(gb[k].astype(np.float64)) + 1e-30, (k, d)
        assert k == "colors_precomp" or np.abs(gb[k]).max() > 0, k          # the colour image is unused: its parameters get exact zeros
    assert np.linalg.norm(ma.astype(np.float64) - mb) <= 1e-4 * np.linalg.norm(mb.astype(np.float64))


@pytest.mark.parametrize("variant", ["ewa", "surfel"])
def test_equal_depths_tie_by_id(variant):
    """Gaussians with bit-identical view depth (copies of one another) inside one tile: the reference's 64-bit (tile | depth) radix sort is stable,
    so they stay in id order.  The per-tile sort ranks on the 32-bit depth key alone and must detect and resolve exactly these collisions."""
    hr = _hiprun()
    W, H, P = 128, 96, 3000
    sc = scenes.make_scene(variant, P, W, H, seed=31)
    m = sc["means3D"].copy()
    m[1000:2000] = m[0:1000]                           # 1000 exact positional duplicates -> equal depth keys
    m[2000:2500, 2] = m[0:500, 2]                      # and 500 with equal z only (equal depth after a pure-rotation-free view? not necessarily: extra mix)
    sc["means3D"] = m
    with oracle.Forward(sc, variant) as f:
        st = hr.run_raw(variant, sc)
        _lists(st, f)
        k = f.keys()
        assert (np.diff(k.astype(np.uint64)) == 0).sum() > 100          # the sorted list really holds runs of equal (tile, depth) keys


@pytest.mark.parametrize("variant,P", [("ewa", 700), ("surfel", 1500), ("plane", 2200), ("surfel", 5000), ("plane", 9000), ("ewa", 40000), ("surfel", 24000)])
def test_long_tile_lists_sort_paths(variant, P, monkeypatch):
    """(GSR_DEPTH_ORDER=tile unless the environment already chose: "auto" would send these gaussian counts to the global sort.)  A 48x32 image (6 tiles) with thousands of gaussians per tile: the per-tile depth sort's paths -- rank by counting (<= 256 entries), the LDS
    bitonic network (fused: <= 1024 entries EWA / 2048 PLANE, SURFEL, the forward's staging LDS; kernel: one wave <= 1024, the workgroup <= 4096) and
    the global-memory radix fallback (longer) -- give the oracle's list bit for bit, and the
    tile keys survive the fallback's use of their array as scratch."""
    hr = _hiprun()
    if "GSR_DEPTH_ORDER" not in os.environ:
        pytest.skip("run by test_per_tile_depth_sort_forced (the library reads GSR_DEPTH_ORDER once per process)")
    W, H = 48, 32
    sc = scenes.make_scene(variant, P, W, H, seed=21)
    with oracle.Forward(sc, variant) as f:
        st = hr.run_raw(variant, sc)
        view = _lists(st, f)
        ft, nc = f.image_state()
        _ncontrib_close(view["n_contrib"][0], nc[0], st["final_T"][0], ft[0])


@pytest.mark.parametrize("W,H,what", [(80, 48, "15 tiles: an odd count, the last 16-bit counter pair half used"),
                                      (2048, 2048, "16384 tiles: the largest image the bucket sort takes"),
                                      (2064, 2048, "16512 tiles: one tile column more, the two radix passes")])
def test_tile_count_edges_of_the_bucket_sort(W, H, what):
    """The one-pass bucket sort on the tile id packs two tiles' 16-bit counters into an LDS word and applies up to GSR_TB_TILES_MAX = 16384 tiles
    (gs-sr_amd/csrc/gsr_binning.hip): an odd tile count, the limit itself and the first size beyond it (which falls back to the radix passes) give the
    oracle's lists, ranges and images."""
    hr = _hiprun()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    sc = scenes.make_scene("ewa", min(6000, 120 * T), W, H, seed=33)          # <= 155 gaussians per tile: the per-tile depth order (gsr_depth_order_static_rule)
    with oracle.Forward(sc, "ewa") as f:
        st = hr.run_raw("ewa", sc)
        assert st["ranges"].shape[0] == T
        _lists(st, f)
        _img_close(st["color"], f.color)
        assert np.array_equal(st["radii"], f.radii)


# The switches that stay in the product, each read once per process, so every set runs in a child process; orthogonal switches share a child (round 4 spent
# eight child processes on them):
#   A  the reference-shaped paths: GSR_BWD=px (round 1's pixel-parallel backward), GSR_DEPTH_ORDER=global (rounds 1-2's global LSD sort of the gaussians),
#      GSR_TILE_CULL=0 (every tile of every rect emitted: R, tiles_touched, point_list, tile keys and ranges bit-exact against the oracle's, see
#      test_forward_backward_parity), GSR_XCD_REMAP=0 (raster launch order);
#   B  GSR_DEPTH_ORDER=tile forced + GSR_TILE_SORT=kernel (the per-tile sort as its own launch), GSR_TILE_BUCKET=0 (two radix passes on the tile id: real tile
#      keys, so `ranges` is held against keys the bucket sort's debug view would have rebuilt from it), GSR_XCD_REMAP=1 (banded launch order);
#   C  GSR_DEPTH_ORDER=tile forced + GSR_TILE_SORT=fused: the long-list paths of the blend forward's prologue.
SWITCH_SETS = {
    "A": (dict(GSR_BWD="px", GSR_DEPTH_ORDER="global", GSR_TILE_CULL="0", GSR_XCD_REMAP="0"),
          "test_forward_backward_parity or test_edge_cases or test_long_tile_lists or test_speculative_forward or test_full_size_properties"),
    "B": (dict(GSR_DEPTH_ORDER="tile", GSR_TILE_SORT="kernel", GSR_TILE_BUCKET="0", GSR_XCD_REMAP="1"),
          "test_forward_backward_parity or test_edge_cases or test_long_tile_lists or test_equal_depths or test_speculative_forward or test_full_size_properties"),
    "C": (dict(GSR_DEPTH_ORDER="tile", GSR_TILE_SORT="fused"), "test_long_tile_lists or test_equal_depths or test_forward_backward_parity"),
}


@pytest.mark.parametrize("which", sorted(SWITCH_SETS))
def test_kept_switches_keep_the_results(which):
    import subprocess
    import sys
    env, select = SWITCH_SETS[which]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", select, "-p", "no:cacheprovider"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout and "skipped" not in r.stdout.splitlines()[-1]


@pytest.mark.parametrize("P,frac,scale,longest", [(24000, 0.6, 0.12, 1562), (40000, 0.7, 0.05, 6000)])
def test_launch_order_feedback_leaves_results_alone(P, frac, scale, longest):
    """The feedback of the longest tile list (a mapped per-device word the blend forward stores into): forwards that follow a report launch their
    blend workgroups longest-list-first (k_tile_order; lists beyond max(1024, 4 x mean)) and, after a list beyond 6000 entries, take the GLOBAL depth
    order instead of the per-tile one.  Both are decided once per forward and change nothing in the results: every output of the later forwards
    stays bit-identical to the first one's (which ran before any report), and equal to the oracle's integer stages."""
    hr = _hiprun()
    W, H = 256, 256
    sc = scenes.make_scene("surfel", P, W, H, seed=13)
    scenes.concentrate(sc, frac, scale)
    runs = [hr.run_raw("surfel", sc) for _ in range(4)]
    lens = runs[0]["ranges"][:, 1].astype(np.int64) - runs[0]["ranges"][:, 0].astype(np.int64)
    assert lens.max() > max(longest, 20 * P // (16 * 16))            # the scene does trigger the report (and, second case, the depth-order switch)
    for r in runs[1:]:
        for k in ("color", "radii", "point_list", "tile_keys", "ranges", "final_T", "n_contrib"):
            assert np.array_equal(r[k], runs[0][k]), k
    with oracle.Forward(sc, "surfel") as f:
        _lists(runs[-1], f)


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_prefiltered_set_and_a_point_behind_the_camera(variant):
    """GaussianRasterizationSettings.prefiltered=True asserts that every gaussian passes the frustum test; the reference prints "Point is filtered
    although prefiltered is set. This shouldn't happen!" and traps the device when one does not (auxiliary.h:156-160).  Here the forward call
    fails with that message (both the two-stage and the single-call forward), and a scene that keeps the promise renders as with prefiltered=False."""
    from gsrast import rasterize as rz
    hr = _hiprun()
    W, H, P = 160, 112, 1500
    sc = scenes.make_scene(variant, P, W, H, seed=5)
    t = hr.to_dev(sc, "cuda")
    vid = hr.VID[variant]
    rs = hr.settings(variant, t)._replace(prefiltered=True)
    args = (t["means3D"], t.get("shs"), t.get("colors_precomp"), t["opacities"], t.get("scales"), t.get("rotations"), t.get("cov3D_precomp"),
            t.get("all_map") if variant == "plane" else None)
    key = (torch.cuda.current_device(), vid, W, H)
    rz._R_HINT.pop(key, None)
    ok1 = rz.forward(vid, *args, rs)                      # no hint: stage1 + stage2
    ok2 = rz.forward(vid, *args, rs)                      # hint: single call
    ref = rz.forward(vid, *args, rs._replace(prefiltered=False))
    assert ok1[0] == ok2[0] == ref[0]
    assert torch.equal(ok2[1]["color"], ref[1]["color"])
    bad = t["means3D"].clone()
    V = t["viewmatrix"]
    behind = torch.tensor([0.0, 0.0, -1.0], device=bad.device)        # one gaussian one unit behind the camera (camera space) -> world
    bad[7] = (behind - V[3, :3]) @ torch.linalg.inv(V[:3, :3])
    for hint in (False, True):
        if not hint:
            rz._R_HINT.pop(key, None)
        with pytest.raises(RuntimeError, match="prefiltered is set"):
            rz.forward(vid, bad, *args[1:], rs)
    rz.forward(vid, bad, *args[1:], rs._replace(prefiltered=False))   # without the promise the point is simply culled


def test_stage2_refuses_a_geom_arena_without_its_depth_order_record():
    """The depth order of a forward (global sort / per-tile sort: it fixes the layout of offsets / scan_tmp / sorted_idx) is recorded IN the geom arena by
    the preprocess kernel; gsr_forward_stage2 reads it back and fails loudly on an arena that carries none (a side table keyed by the arena's
    address could miss -- after 256 other arenas, or when the caller moved the bytes -- and stage 2 then silently binned on the static rule).  A MOVED copy
    of the arena is accepted: the record travels with the bytes."""
    import ctypes as C
    from gsrast import rasterize as rz, Outputs, lib, ptr, stream_ptr, check
    hr = _hiprun()
    W, H, P = 160, 112, 2000
    sc = scenes.make_scene("surfel", P, W, H, seed=9)
    t = hr.to_dev(sc, "cuda")
    rs = hr.settings("surfel", t)
    vid = hr.VID["surfel"]
    rz._R_HINT.pop((torch.cuda.current_device(), vid, W, H), None)
    R, outs, radii, geom, binning, img = rz.forward(vid, t["means3D"], None, t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], None, None, rs)
    L = lib()
    cfg, inp, keep, _, _ = rz._prepare(vid, t["means3D"], None, t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], None, None, rs)
    color2, others2 = torch.empty_like(outs["color"]), torch.empty_like(outs["others"])
    o = Outputs(ptr(color2), ptr(others2), None, None, None)
    bin2 = torch.empty_like(binning)
    moved = geom.clone()                                   # the arena's bytes at another address: stage 2 must find the record in them
    check(L.gsr_forward_stage2(C.byref(cfg), C.byref(inp), ptr(moved), moved.numel(), ptr(bin2), bin2.numel(), ptr(img), img.numel(), R, C.byref(o),
                               stream_ptr(geom.device)), "forward")
    torch.cuda.synchronize()
    assert torch.equal(color2, outs["color"]) and torch.equal(others2, outs["others"])
    # ABI 7: the word stage 1 returned, handed back -> stage 2 does not touch the host at all; same results
    order = C.c_uint32(0); R1 = C.c_uint32(0)
    radii2 = torch.empty_like(radii)
    check(L.gsr_forward_stage1_ex(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(radii2), C.byref(R1), C.byref(order), stream_ptr(geom.device)), "forward")
    assert R1.value == R and order.value != 0 and torch.equal(radii2, radii)
    color2.zero_(); others2.zero_()
    check(L.gsr_forward_stage2_ex(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(bin2), bin2.numel(), ptr(img), img.numel(), R, order.value, C.byref(o),
                                  stream_ptr(geom.device)), "forward")
    torch.cuda.synchronize()
    assert torch.equal(color2, outs["color"]) and torch.equal(others2, outs["others"])
    # cfg->debug: a word that is not the record stage 1 left (here: the other order's) is refused instead of binning the arena in the wrong layout
    other = 0x47530001 if order.value == 0x47530002 else 0x47530002
    cfg.debug = 1
    with pytest.raises(RuntimeError, match="not the record stage 1 left"):
        check(L.gsr_forward_stage2_ex(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(bin2), bin2.numel(), ptr(img), img.numel(), R, other, C.byref(o),
                                      stream_ptr(geom.device)), "forward")
    check(L.gsr_forward_stage2_ex(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(bin2), bin2.numel(), ptr(img), img.numel(), R, order.value, C.byref(o),
                                  stream_ptr(geom.device)), "forward")
    cfg.debug = 0
    blank = torch.zeros_like(geom)                         # never seen by a preprocess kernel
    with pytest.raises(RuntimeError, match="depth-order record"):
        check(L.gsr_forward_stage2(C.byref(cfg), C.byref(inp), ptr(blank), blank.numel(), ptr(bin2), bin2.numel(), ptr(img), img.numel(), R, C.byref(o),
                                   stream_ptr(geom.device)), "forward")
