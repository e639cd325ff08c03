"""CPU tests that pin the oracle (oracle/gsr_oracle.c): golden vectors from the reference's Python helpers, float64
autograd of an independent forward restatement, invariants and the fork-specific quirks (SURVEY.md §7 hard parts)."""
import os

import numpy as np
import pytest

import oracle
import ref_torch
import scenes


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


# ------------------------------------------------------------------ golden vectors (reference python helpers)
@pytest.mark.parametrize("i", [0, 1, 2])
def test_camera_conventions_match_reference_helpers(golden_dir, i):
    g = np.load(os.path.join(golden_dir, f"camera_{i}.npz"))
    wvt = scenes.world2view(g["R"], g["T"]).T
    proj = scenes.projection_matrix(0.01, 100.0, float(g["fovx"]), float(g["fovy"])).T
    full = wvt.astype(np.float32) @ proj.astype(np.float32)
    assert np.allclose(wvt, g["world_view_transform"], atol=1e-6)
    assert np.allclose(proj, g["projection_matrix"], atol=1e-6)
    assert np.allclose(full, g["full_proj_transform"], atol=1e-5)
    center = np.linalg.inv(wvt.astype(np.float64))[3, :3]
    assert np.allclose(center, g["camera_center"], atol=1e-5)
    assert abs(float(g["focal_back"]) - float(g["fx"])) < 1e-3


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_matches_reference_eval_sh(golden_dir, deg):
    g = np.load(os.path.join(golden_dir, "sh_eval.npz"))
    sh, dirs = g["sh"], g["dirs"]                      # sh: [P,3,16] (eval_sh layout)
    P = sh.shape[0]
    sc = scenes.make_scene("ewa", P, 64, 48, seed=0, color_mode="sh", sh_degree=deg)
    campos = np.array([0.3, -0.2, 5.0], np.float32)    # campos is an independent kernel input
    sc["campos"] = campos
    sc["means3D"] = (campos[None] + dirs).astype(np.float32)
    sc["shs"] = np.ascontiguousarray(sh.transpose(0, 2, 1)).astype(np.float32)   # rasterizer layout [P,16,3]
    sc["scales"] = np.full((P, 3), 0.05, np.float32)
    with oracle.Forward(sc, "ewa") as f:
        assert (f.radii > 0).all()
        rgb = f.geom()["rgb"]
    # the direction the rasterizer uses is normalise(mean - campos) in float32
    expect = np.maximum(g[f"deg{deg}"] + 0.5, 0.0)
    assert np.abs(rgb - expect).max() < 2e-5


# ------------------------------------------------------------------ oracle backward vs float64 autograd
CASES = [("ewa", "precomp"), ("ewa", "sh"), ("plane", "precomp"), ("surfel", "precomp"), ("surfel", "sh")]


@pytest.mark.parametrize("variant,cm", CASES)
def test_oracle_matches_float64_autograd(variant, cm):
    W, H, P = 64, 48, 250
    sc = scenes.make_scene(variant, P, W, H, seed=3, color_mode=cm, sigma_px=3.0, bg=(0.3, 0.5, 0.7), pose=1)
    og = scenes.random_out_grads(variant, W, H, seed=3, scale=1.0)
    if variant == "surfel":
        og["dL_dothers"][8:11] = 0          # median-normal quirk is tested separately
    with oracle.Forward(sc, variant) as f:
        g = f.backward(**og)
        out, tg = ref_torch.backward(variant, sc, og)
        assert np.array_equal(f.radii, out["radii"].numpy())
        assert np.abs(f.color - out["color"].detach().numpy()).max() < 2e-5
        if variant == "surfel":
            oth = out["others"].detach().numpy()
            for ch in range(11):
                assert np.abs(f.others[ch] - oth[ch]).max() <= 2e-5 * max(1.0, np.abs(oth[ch]).max()), ch
        if variant == "plane":
            assert np.array_equal(f.observe, out["observe"].numpy())
            assert _rel(f.out_all_map, out["out_all_map"].detach().numpy()) < 1e-5
            assert _rel(f.plane_depth, out["plane_depth"].detach().numpy()) < 1e-5
    tol = 5e-4 if variant == "plane" else 1e-4     # float32 oracle vs float64 autograd; target for gradients is 1e-3
    assert _rel(g["dL_dmeans3D"], tg["means3D"]) < tol
    assert _rel(g["dL_dscales"], tg["scales"]) < tol
    assert _rel(g["dL_drotations"], tg["rotations"]) < tol
    assert _rel(g["dL_dopacity"][:, 0], tg["opacities"].reshape(-1)) < tol
    if cm == "sh":
        assert _rel(g["dL_dsh"], tg["shs"]) < tol
    else:
        assert _rel(g["dL_dcolors"], tg["colors_precomp"]) < tol
    if variant != "surfel":
        assert _rel(g["dL_dmeans2D"], tg["means2D"]) < tol
    if variant == "plane":
        assert _rel(g["dL_dall_map"], tg["all_map"]) < tol


def test_plane_abs_gradient_matches_per_pair_autograd():
    """PLANE backward.cu:602-603: means2D_abs accumulates |per-(pixel,gaussian) gradient|."""
    W, H, P = 48, 32, 120
    sc = scenes.make_scene("plane", P, W, H, seed=5, sigma_px=3.0)
    og = scenes.random_out_grads("plane", W, H, seed=5, scale=1.0)
    with oracle.Forward(sc, "plane") as f:
        g = f.backward(**og)
    out, tg = ref_torch.backward("plane", sc, og, pair_xy_leaf=True)
    order = out["_order"].numpy()
    pg = tg["_pair_xy"]                                     # [Npix, G, 2] d L / d (pixel-space mean), per pair
    expect = np.zeros((P, 2))
    expect[order, 0] = np.abs(pg[:, :, 0]).sum(0) * 0.5 * W
    expect[order, 1] = np.abs(pg[:, :, 1]).sum(0) * 0.5 * H
    assert _rel(g["dL_dmeans2D_abs"][:, :2], expect) < 5e-5


# ------------------------------------------------------------------ invariants
@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_binning_invariants(variant):
    sc = scenes.make_scene(variant, 3000, 160, 112, seed=2)
    with oracle.Forward(sc, variant) as f:
        keys, pl, rng, tt = f.keys(), f.point_list(), f.ranges(), f.tiles_touched()
        assert f.R == int(tt.sum()) and f.R > 0
        assert np.all(np.diff(keys.astype(np.uint64)) >= 0)                 # sorted by (tile, depth)
        tiles = (keys >> np.uint64(32)).astype(np.int64)
        for t in np.unique(tiles):
            idx = np.nonzero(tiles == t)[0]
            assert rng[t, 0] == idx[0] and rng[t, 1] == idx[-1] + 1
        untouched = np.setdiff1d(np.arange(f.T), np.unique(tiles))
        assert np.all(rng[untouched] == 0)
        # ties keep gaussian-index order (stable sort)
        same = np.nonzero(np.diff(keys.astype(np.uint64)) == 0)[0]
        assert np.all(pl[same] < pl[same + 1])
        ft, nc = f.image_state()
        assert (ft[0] <= 1.0).all() and (ft[0] >= 0.0).all()
        if variant == "surfel":
            assert np.allclose(f.others[1], 1 - ft[0], atol=1e-6)           # alpha map = 1 - final_T
            assert (nc[1] <= nc[0]).all()                                   # median contributor <= last contributor


def test_single_isotropic_gaussian_closed_form():
    """One isotropic splat at the image centre: alpha, colour follow the closed form (3DGS forward.cu:333-357)."""
    W, H = 64, 48
    cam = scenes.make_camera(W, H, 60.0, 60.0)
    z, s, o = 4.0, 0.2, 0.8
    sc = dict(cam)
    sc.update(means3D=np.array([[0, 0, z]], np.float32), scales=np.full((1, 3), s, np.float32),
              rotations=np.array([[1, 0, 0, 0]], np.float32), opacities=np.array([[o]], np.float32),
              colors_precomp=np.array([[0.2, 0.5, 0.9]], np.float32), bg=np.array([0.1, 0.1, 0.1], np.float32),
              scale_modifier=1.0, sh_degree=0)
    with oracle.Forward(sc, "ewa") as f:
        var = (60.0 * s / z) ** 2 + 0.3
        cx, cy = (W - 1) / 2.0, (H - 1) / 2.0        # ndc 0 -> ((0+1)*W-1)/2
        ys, xs = np.mgrid[0:H, 0:W]
        power = -0.5 * ((xs - cx) ** 2 + (ys - cy) ** 2) / var
        alpha = np.minimum(0.99, o * np.exp(power))
        alpha[alpha < 1 / 255.0] = 0
        r = int(f.radii[0])
        assert r == int(np.ceil(3 * np.sqrt(var)))
        tile_ok = np.zeros((H, W), bool)
        x0, x1 = max(0, int((cx - r) / 16)), min(4, int((cx + r + 15) / 16))
        y0, y1 = max(0, int((cy - r) / 16)), min(3, int((cy + r + 15) / 16))
        tile_ok[y0 * 16:y1 * 16, x0 * 16:x1 * 16] = True
        alpha[~tile_ok] = 0
        for ch, c in enumerate([0.2, 0.5, 0.9]):
            assert np.abs(f.color[ch] - (alpha * c + (1 - alpha) * 0.1)).max() < 1e-5


# ------------------------------------------------------------------ fork-specific quirks
def test_surfel_median_normal_gradient_goes_to_every_contributor():
    """SURFEL backward.cu:381."""
    W, H, P = 32, 32, 60
    sc = scenes.make_scene("surfel", P, W, H, seed=9, sigma_px=5.0)
    og = dict(dL_dcolor=np.zeros((3, H, W), np.float32), dL_dothers=np.zeros((11, H, W), np.float32))
    og["dL_dothers"][8] = 1.0
    with oracle.Forward(sc, "surfel") as f:
        g = f.backward(**og)
        ft, nc = f.image_state()
        pl, rng = f.point_list(), f.ranges()
    # dL_dnormal[:,0] counts, per gaussian, the number of (pixel, gaussian) pairs that contributed
    assert g["dL_dconic"][:, 0].sum() > 0
    assert np.allclose(g["dL_dconic"][:, 0], np.round(g["dL_dconic"][:, 0]))
    assert np.abs(g["dL_dconic"][:, 1:]).max() == 0


def test_surfel_means2D_is_densification_proxy():
    """SURFEL backward.cu:633-636: dL_dmean2D <- dL_dtransMat[2|5] * depth * 0.5 * (W|H)."""
    W, H, P = 64, 48, 200
    sc = scenes.make_scene("surfel", P, W, H, seed=4)
    og = scenes.random_out_grads("surfel", W, H, seed=4, scale=1.0)
    with oracle.Forward(sc, "surfel") as f:
        g = f.backward(**og)
        T = f.geom()["cov"]
        vis = f.radii > 0
    assert np.allclose(g["dL_dmeans2D"][vis, 0], g["dL_dcov3D"][vis, 2] * T[vis, 8] * 0.5 * W, rtol=1e-5, atol=1e-7)
    assert np.allclose(g["dL_dmeans2D"][vis, 1], g["dL_dcov3D"][vis, 5] * T[vis, 8] * 0.5 * H, rtol=1e-5, atol=1e-7)
    assert np.abs(g["dL_dmeans2D"][~vis]).max() == 0 if (~vis).any() else True


def test_ewa_quaternion_is_not_normalised():
    """3DGS forward.cu:127: scaling the quaternion by k scales R's off-diagonal terms -> different image."""
    sc = scenes.make_scene("ewa", 200, 64, 48, seed=6)
    with oracle.Forward(sc, "ewa") as f0:
        c0 = f0.color.copy()
    sc2 = dict(sc); sc2["rotations"] = sc["rotations"] * 1.5
    with oracle.Forward(sc2, "ewa") as f1:
        assert np.abs(f1.color - c0).max() > 1e-3
    scs = scenes.make_scene("surfel", 200, 64, 48, seed=6)
    with oracle.Forward(scs, "surfel") as f0:
        c0 = f0.color.copy()
    scs2 = dict(scs); scs2["rotations"] = scs["rotations"] * 1.5      # SURFEL normalises (auxiliary.h:215-223)
    with oracle.Forward(scs2, "surfel") as f1:
        assert np.abs(f1.color - c0).max() < 1e-5


def test_visible_filter_equals_ewa_radii():
    """FILTER forward.cu:268-340 is the EWA preprocess truncated to radii."""
    sc = scenes.make_scene("ewa", 5000, 320, 240, seed=8)
    with oracle.Forward(sc, "ewa") as f:
        assert np.array_equal(oracle.visible_filter(sc), f.radii)
    mv = oracle.mark_visible(sc["means3D"], sc["viewmatrix"], sc["projmatrix"])
    pv = sc["means3D"] @ sc["viewmatrix"][:3, :3] + sc["viewmatrix"][3, :3]
    assert np.array_equal(mv, pv[:, 2] > 0.2)
