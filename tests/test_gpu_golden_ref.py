"""The HIP ops (through the C-ABI) against vectors produced by running the reference's own Python
(tests/golden/ref_*.npz, generator tests/golden/make_golden_ref.py).  Tolerances: north_star's 1e-4 on values, 1e-3 on gradients
(relative to the largest reference entry), tighter where float32 allows."""
import numpy as np
import pytest
import torch

import golden_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("name", golden_ref.DECODE)
def test_decode_matches_reference_run(name):
    from test_gpu_decode import _run_hip
    case, dL, exp, grads = golden_ref.decode_case(name)
    h, g = _run_hip(case, dL)
    assert np.array_equal(h["mask"].astype(bool), exp["mask"])
    np.testing.assert_allclose(h["neural_opacity"].reshape(-1), exp["neural_opacity"], rtol=1e-5, atol=2e-6)
    for n in ("xyz", "color", "scaling", "rot"):
        np.testing.assert_allclose(h[n], exp[n], rtol=1e-5, atol=2e-6, err_msg=n)
    np.testing.assert_allclose(h["opacity"].reshape(-1), exp["opacity"].reshape(-1), rtol=1e-5, atol=2e-6)
    assert set(g) == set(grads)
    for n, r in grads.items():
        assert _rel(g[n].reshape(r.shape), r) < 1e-4, (n, _rel(g[n].reshape(r.shape), r))


@pytest.mark.parametrize("mode", ["floor", "round", "ceil", "progressive"])
def test_octree_mask_matches_reference_run(mode):
    """gsr_octree_visible's level-of-detail half: `anchor_mask` / `prog_ratio` / `transition_mask` do not depend on the frustum, so any
    camera placed at the fixture's camera centre serves."""
    import scaffold_filter
    import scenes
    from gsrast import octree
    z = golden_ref.load(f"ref_lod_{mode}")
    Na = z["anchor"].shape[0]
    cam = scenes.make_camera(320, 240, 256.0, 256.0, yaw_deg=0.0, t=(0.0, 0.0, 0.0))
    t = lambda a, **k: torch.tensor(np.ascontiguousarray(a), device=DEV, **k)
    fs = scaffold_filter.GaussianRasterizationSettings(
        image_height=240, image_width=320, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=DEV), scale_modifier=1.0,
        viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=0, campos=t(z["campos"]), prefiltered=False, debug=False)
    rot = torch.zeros(Na, 4, device=DEV); rot[:, 0] = 1
    out = octree.octree_visible(fs, t(z["anchor"]), t(z["level"]).unsqueeze(1), torch.full((Na, 6), 0.05, device=DEV), rot, float(z["voxel_size"]),
                                float(z["fork"]), float(z["standard_dist"]), int(z["levels"]), dist2level=mode, extra_level=t(z["extra_level"]),
                                resolution_scale=float(z["resolution_scale"]))
    diff = out["anchor_mask"].cpu().numpy() != z["anchor_mask"]
    assert diff.sum() <= 2, int(diff.sum())                                   # float32 log2 on a level boundary
    if mode == "progressive":
        same = ~diff & (out["transition_mask"].cpu().numpy() == z["transition_mask"])
        assert (~same).sum() <= 4
        np.testing.assert_allclose(out["prog_ratio"].cpu().numpy().reshape(-1)[same], z["prog_ratio"][same], rtol=0, atol=1e-4)


def test_l1_ssim_matches_reference_run():
    from gsrast.losses import l1_ssim
    z = golden_ref.load("ref_loss_l1_ssim")
    lam = float(z["lambda_dssim"])
    x = torch.tensor(z["image"], device=DEV, requires_grad=True)
    loss, parts = l1_ssim(x, torch.tensor(z["gt"], device=DEV), lam, return_parts=True)
    loss.backward()
    np.testing.assert_allclose((1 - lam) * parts[0].item(), float(z["L1_loss"]), rtol=2e-5)
    np.testing.assert_allclose(lam * (1 - parts[1].item()), float(z["ssim_loss"]), rtol=2e-5)
    np.testing.assert_allclose(loss.item(), float(z["total"]), rtol=2e-5)
    assert _rel(x.grad.cpu().numpy(), z["d_image"]) < 1e-4


@pytest.mark.parametrize("ratio", [0, 1])
def test_surfel_geo_matches_reference_run(ratio):
    from gsrast.losses import camera_ray_matrices, surfel_geo_loss
    z = golden_ref.load(f"ref_loss_surfel_geo_r{ratio}")
    W, H = int(z["W"]), int(z["H"])
    rm, nr = camera_ray_matrices(torch.tensor(z["viewmatrix"], device=DEV), torch.tensor(z["projmatrix"], device=DEV), W, H)
    am = np.zeros((11, H, W), np.float32); am[:7] = z["allmap"]
    x = torch.tensor(am, device=DEV, requires_grad=True)
    ln, ld = float(z["lambda_normal"]), float(z["lambda_dist"])
    loss, parts, depth, nw, sn = surfel_geo_loss(x, rm, nr, float(z["depth_ratio"]), ln, ld, return_maps=True)
    loss.backward()
    np.testing.assert_allclose(ln * parts[0].item(), float(z["normal_loss"]), rtol=1e-4)
    np.testing.assert_allclose(ld * parts[1].item(), float(z["dist_loss"]), rtol=1e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), z["surf_depth"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(nw.cpu().numpy(), z["normal"], rtol=0, atol=1e-5)
    snr = z["surf_normal"]
    assert np.linalg.norm(sn.cpu().numpy() - snr) / np.linalg.norm(snr) < 2e-4
    g = x.grad.cpu().numpy()
    assert not g[7:].any() and not g[:7][z["d_allmap_nan"]].any()
    r = z["d_allmap"]
    assert np.abs(g[:7] - r).max() <= 5e-3 * np.abs(r).max()
    assert np.linalg.norm(g[:7] - r) / np.linalg.norm(r) < 1e-3


def test_plane_geo_matches_reference_run():
    from gsrast.losses import plane_geo_loss
    z = golden_ref.load("ref_loss_plane_geo")
    K = np.array([[z["fx"], 0, z["cx"]], [0, z["fy"], z["cy"]], [0, 0, 1]], np.float64)
    rm = torch.tensor(np.linalg.inv(K.T), dtype=torch.float32, device=DEV)
    d = torch.tensor(z["plane_depth"], device=DEV, requires_grad=True)
    a = torch.tensor(z["out_all_map"], device=DEV, requires_grad=True)
    loss, part, dn = plane_geo_loss(d, a, rm, torch.tensor(z["image_weight"], device=DEV), float(z["lambda_normal"]), return_map=True)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["normal_loss"]), rtol=1e-4)
    dnr = z["depth_normal"]
    assert np.linalg.norm(dn.cpu().numpy() - dnr) / np.linalg.norm(dnr) < 2e-4
    gd = z["d_plane_depth"]
    assert np.linalg.norm(d.grad.cpu().numpy() - gd) / np.linalg.norm(gd) < 1e-3
    ga = z["d_out_all_map"]
    assert np.abs(a.grad.cpu().numpy() - ga).max() <= 1e-5 * np.abs(ga).max() + 1e-12


def test_tsdf_matches_reference_run():
    """gsr_tsdf_integrate vs gssr/utils/mesh_utils.py compute_unbounded_tsdf (3 frames; adaptive truncation pass + texturing pass)."""
    from gsrast.tsdf import tsdf_integrate_
    z = golden_ref.load("ref_tsdf_unbounded")
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)
    for pts, trunc, key, exp in ((z["points"], t(z["sdf_trunc"]), "tsdf", z["tsdf"]), (z["verts"], 5 * float(z["voxel_size"]), "rgb", z["vert_rgb"])):
        V = pts.shape[0]
        tsdf = torch.ones(V, device=DEV); w = torch.ones(V, device=DEV); rgb = torch.zeros(V, 3, device=DEV)
        for F, d, c in zip(z["full_proj"], z["depth"], z["rgb"]):
            tsdf_integrate_(t(pts), t(F), t(d), t(c), trunc, tsdf, rgb, w)
        got = {"tsdf": tsdf, "rgb": rgb}[key].cpu().numpy()
        assert np.abs(got - exp).max() < 1e-4, (key, np.abs(got - exp).max())
        assert (w > 1).float().mean().item() > 0.2


def test_plane_allmap_matches_reference_run_and_oracle():
    """gsr_plane_allmap[_backward] vs PGSRScene.render()'s captured all_map + autograd (fixture), and vs the oracle at 300k Gaussians."""
    import oracle_multiview as om
    from gsrast.plane_prep import plane_input_all_map
    z = golden_ref.load("ref_plane_allmap")
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)
    x = t(z["means3D"]).requires_grad_(True); q = t(z["rotations"]).requires_grad_(True); s = t(z["scales"]).requires_grad_(True)
    am = plane_input_all_map(x, q, s, t(z["viewmatrix"]), t(z["campos"]))
    (am * t(z["dL_dall_map"])).sum().backward()
    np.testing.assert_allclose(am.detach().cpu().numpy(), z["all_map"], rtol=2e-5, atol=2e-6)
    assert s.grad is None
    assert np.abs(x.grad.cpu().numpy() - z["d_means3D"]).max() <= 2e-5 * np.abs(z["d_means3D"]).max()
    assert np.abs(q.grad.cpu().numpy() - z["d_rotations"]).max() <= 5e-5 * np.abs(z["d_rotations"]).max()
    # full size, get_scaling-shaped (P,6) scales, against the oracle
    r = np.random.default_rng(2)
    P = 300000
    xyz = r.uniform(-5, 5, (P, 3)).astype(np.float32); qq = r.normal(0, 1, (P, 4)).astype(np.float32)
    sc = np.exp(r.normal(-2, 0.7, (P, 6))).astype(np.float32); dL = r.normal(0, 1, (P, 5)).astype(np.float32)
    oam, odx, odq = om.plane_allmap(xyz, qq, sc[:, :3], z["viewmatrix"], z["campos"], dL)
    x = t(xyz).requires_grad_(True); q = t(qq).requires_grad_(True)
    am = plane_input_all_map(x, q, t(sc), t(z["viewmatrix"]), t(z["campos"]))
    (am * t(dL)).sum().backward()
    np.testing.assert_allclose(am.detach().cpu().numpy(), oam, rtol=1e-5, atol=1e-6)
    assert np.abs(x.grad.cpu().numpy() - odx).max() <= 1e-5 * np.abs(odx).max()
    assert np.abs(q.grad.cpu().numpy() - odq).max() <= 1e-5 * np.abs(odq).max()


def test_training_stats_matches_reference_run_and_oracle():
    """gsd_training_stats vs ScaffoldGaussian.training_statis (fixture: two consecutive calls on the reference's model object) and vs the
    oracle at 60k anchors x 10 offsets."""
    import oracle_decode
    from gsrast.decode import training_stats_
    from test_golden_ref_cpu import _stats_fixture
    z, names, calls = _stats_fixture()
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)
    acc = {n: t(z["before_" + n]) for n in names}
    for c in calls:
        training_stats_(acc["opacity_accum"], acc["anchor_demon"], acc["offset_gradient_accum"], acc["offset_denom"], t(c["grad"]),
                        t(c["neural_opacity"]), t(c["update_filter"]), t(c["selection"]), anchor_visible_mask=t(c["visible"]))
    for n in names:
        np.testing.assert_allclose(acc[n].cpu().numpy(), z["after_" + n], rtol=1e-6, atol=1e-6, err_msg=n)
    r = np.random.default_rng(4)
    Na, k = 60000, 10
    vis = r.uniform(size=Na) < 0.55; Nv = int(vis.sum())
    nop = np.tanh(r.normal(0, 1, Nv * k)).astype(np.float32); sel = nop > 0; P = int(sel.sum())
    upd = r.uniform(size=P) < 0.8; grad = r.normal(0, 1, (P, 3)).astype(np.float32)
    host = [r.uniform(0, 1, Na).astype(np.float32), r.integers(0, 9, Na).astype(np.float32), r.uniform(0, 1, Na * k).astype(np.float32),
            r.integers(0, 9, Na * k).astype(np.float32)]
    devt = [t(a.reshape(-1, 1)) for a in host]
    oracle_decode.training_stats(np.nonzero(vis)[0], k, nop, sel, upd, grad, *host)
    training_stats_(*devt, t(grad), t(nop), t(upd), t(sel), anchor_visible_mask=t(vis))
    for a, b in zip(host, devt):
        assert np.array_equal(a, b.cpu().numpy().reshape(-1))                # same float operations in the same order: bit-exact


def test_densify_stats_matches_reference_run():
    """gsr_densify_stats vs the reference's VanillaGaussian.densify / PGSRGaussian.densify statistics (fixture)."""
    from gsrast.stats import densification_stats_
    z = golden_ref.load("ref_densify_stats")
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)
    for tag, names in (("vanilla", ("max_radii2D", "xyz_gradient_accum", "denom")),
                       ("pgsr", ("max_radii2D", "xyz_gradient_accum", "denom", "xyz_gradient_accum_abs", "denom_abs"))):
        a = {n: t(z["init_" + n]) for n in names}
        if tag == "vanilla":
            densification_stats_(a["max_radii2D"], a["xyz_gradient_accum"], a["denom"], t(z["grad"]), t(z["visibility_filter"]), t(z["radii"]))
        else:
            densification_stats_(a["max_radii2D"], a["xyz_gradient_accum"], a["denom"], t(z["grad"]), t(z["visibility_filter"]), t(z["radii"]),
                                 t(z["out_observe"]), t(z["grad_abs"]), a["xyz_gradient_accum_abs"], a["denom_abs"])
        for n in names:
            np.testing.assert_allclose(a[n].cpu().numpy(), z[f"{tag}_{n}"], rtol=1e-6, atol=1e-6, err_msg=f"{tag}:{n}")


def test_cov3d_precomp_matches_reference_python_twin():
    """diff_gaussian_rasterization with scales + rotations vs with cov3D_precomp = the covariance the reference's own Python helpers build
    (fixture ref_cov3d.npz): same radii and image; also vs the oracle on the precomp path."""
    import hiprun
    import oracle
    from test_golden_ref_cpu import _cov_scene
    z, a, b = _cov_scene()
    ra = hiprun.run("ewa", a, device=DEV); rb = hiprun.run("ewa", b, device=DEV)
    assert np.array_equal(ra["radii"], rb["radii"]) and (ra["radii"] > 0).sum() > 50
    np.testing.assert_allclose(ra["color"], rb["color"], rtol=0, atol=2e-5)
    with oracle.Forward(b, "ewa") as fb:
        assert np.array_equal(rb["radii"], fb.radii)
        np.testing.assert_allclose(rb["color"], fb.color, rtol=0, atol=1e-4)
