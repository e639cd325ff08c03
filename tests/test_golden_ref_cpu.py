"""The C oracles against vectors produced by the reference's own Python (tests/golden/make_golden_ref.py ran
gssr/scene/*_scene.py and gssr/gaussian/*_gaussian.py on CPU torch, float32).  This is what pins oracle/gsd_oracle.c and
oracle/gsl_oracle.c to the reference rather than to our transcription of it."""
import numpy as np
import pytest

import golden_ref
import oracle
import oracle_decode


def _rel(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("name", golden_ref.DECODE)
def test_decode_oracle_matches_reference_run(name):
    case, dL, exp, grads = golden_ref.decode_case(name)
    o = oracle_decode.forward(case)
    assert np.array_equal(o["mask"].astype(bool), exp["mask"])                       # gate margin of the fixtures is > 1e-4
    assert o["P"] == exp["xyz"].shape[0]
    np.testing.assert_allclose(o["neural_opacity"], exp["neural_opacity"], rtol=1e-5, atol=2e-6)
    for n in ("xyz", "color", "scaling", "rot"):
        np.testing.assert_allclose(o[n], exp[n], rtol=1e-5, atol=2e-6, err_msg=n)
    np.testing.assert_allclose(o["opacity"], exp["opacity"].reshape(-1), rtol=1e-5, atol=2e-6)
    g = oracle_decode.backward(case, o["mask"], dL)
    assert set(g) == set(grads)
    for n, r in grads.items():
        assert _rel(g[n].reshape(r.shape), r) < 2e-5, (n, _rel(g[n].reshape(r.shape), r))    # both float32; different summation order


@pytest.mark.parametrize("mode", ["floor", "round", "ceil", "progressive"])
def test_lod_mask_oracle_matches_reference_run(mode):
    z = golden_ref.load(f"ref_lod_{mode}")
    m, pr, tr = oracle_decode.lod_mask(z["anchor"], z["level"], z["extra_level"], z["campos"], float(z["voxel_size"]), float(z["fork"]),
                                       float(z["standard_dist"]), float(z["resolution_scale"]), int(z["levels"]),
                                       ["floor", "round", "ceil", "progressive"].index(mode))
    # integer levels come from floor/round/ceil of a float32 log2: an anchor sitting within 1 ulp of a level boundary may differ
    diff = m != z["anchor_mask"]
    assert diff.sum() <= 2, int(diff.sum())
    if mode == "progressive":
        ok = ~diff
        assert (tr != z["transition_mask"]).sum() <= 2
        same = ok & (tr == z["transition_mask"])
        np.testing.assert_allclose(pr[same], z["prog_ratio"][same], rtol=0, atol=2e-5)


def test_l1_ssim_oracle_matches_reference_run():
    z = golden_ref.load("ref_loss_l1_ssim")
    lam = float(z["lambda_dssim"])
    out, d = oracle.loss_l1_ssim(z["image"], z["gt"], lam)
    # out = {mean |x-gt|, ssim, (1-lam) l1 + lam (1-ssim)}
    np.testing.assert_allclose((1 - lam) * out[0], float(z["L1_loss"]), rtol=2e-5)
    np.testing.assert_allclose(lam * (1 - out[1]), float(z["ssim_loss"]), rtol=2e-5)
    np.testing.assert_allclose(out[2], float(z["total"]), rtol=2e-5)
    assert _rel(d, z["d_image"]) < 5e-5


@pytest.mark.parametrize("ratio", [0, 1])
def test_surfel_geo_oracle_matches_reference_run(ratio):
    import torch
    import ref_geo_torch
    z = golden_ref.load(f"ref_loss_surfel_geo_r{ratio}")
    W, H = int(z["W"]), int(z["H"])
    rm, nr = ref_geo_torch.ray_matrices(torch.tensor(z["viewmatrix"], dtype=torch.float64), torch.tensor(z["projmatrix"], dtype=torch.float64), W, H)
    am = np.zeros((11, H, W), np.float32); am[:7] = z["allmap"]     # our rasterizer's allmap carries 4 more channels; the reference's has 7
    o = oracle.loss_surfel_geo(am, rm.numpy(), nr.numpy(), float(z["depth_ratio"]), float(z["lambda_normal"]), float(z["lambda_dist"]))
    # loss = {mean normal_error, mean rend_dist, total}
    np.testing.assert_allclose(float(z["lambda_normal"]) * o["loss"][0], float(z["normal_loss"]), rtol=5e-5)
    np.testing.assert_allclose(float(z["lambda_dist"]) * o["loss"][1], float(z["dist_loss"]), rtol=5e-5)
    np.testing.assert_allclose(o["loss"][2], float(z["normal_loss"]) + float(z["dist_loss"]), rtol=5e-5)
    assert not o["dL_dallmap"][7:].any()
    o["dL_dallmap"] = o["dL_dallmap"][:7]
    np.testing.assert_allclose(o["surf_depth"], z["surf_depth"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o["normal_world"], z["normal"], rtol=0, atol=1e-5)
    # unit normals from float32 cross products of nearly parallel differences: compare loosely per pixel, tightly in norm
    np.testing.assert_allclose(o["surf_normal"], z["surf_normal"], rtol=0, atol=2e-3)
    assert np.linalg.norm(o["surf_normal"] - z["surf_normal"]) / np.linalg.norm(z["surf_normal"]) < 2e-4
    g = z["d_allmap"]
    assert not o["dL_dallmap"][z["d_allmap_nan"]].any()            # where autograd produced NaN (alpha == 0) the restatement writes 0
    assert np.abs(o["dL_dallmap"] - g).max() <= 5e-3 * np.abs(g).max()
    assert np.linalg.norm(o["dL_dallmap"] - g) / np.linalg.norm(g) < 5e-4


def test_plane_geo_oracle_matches_reference_run():
    z = golden_ref.load("ref_loss_plane_geo")
    K = np.array([[z["fx"], 0, z["cx"]], [0, z["fy"], z["cy"]], [0, 0, 1]], np.float64)
    rm = np.linalg.inv(K.T)
    oam = z["out_all_map"]
    o = oracle.loss_plane_geo(z["plane_depth"][0], oam[3], oam[0:3], z["image_weight"], rm, float(z["lambda_normal"]))
    np.testing.assert_allclose(o["loss"][2], float(z["normal_loss"]), rtol=5e-5)
    np.testing.assert_allclose(o["depth_normal"], z["depth_normal"], rtol=0, atol=2e-3)
    assert np.linalg.norm(o["depth_normal"] - z["depth_normal"]) / np.linalg.norm(z["depth_normal"]) < 2e-4
    gd = z["d_plane_depth"][0]
    assert np.abs(o["dL_ddepth"] - gd).max() <= 5e-3 * np.abs(gd).max()
    assert np.linalg.norm(o["dL_ddepth"] - gd) / np.linalg.norm(gd) < 5e-4
    ga = z["d_out_all_map"]
    assert not ga[3:].any()
    assert np.abs(o["dL_dnormal"] - ga[0:3]).max() <= 1e-5 * np.abs(ga).max() + 1e-12


def test_tsdf_oracle_matches_reference_run():
    """ref_tsdf_integrate vs gssr/utils/mesh_utils.py compute_unbounded_tsdf driven through extract_mesh_unbounded (3 frames)."""
    z = golden_ref.load("ref_tsdf_unbounded")
    V = z["points"].shape[0]
    tsdf = np.ones(V, np.float32); w = np.ones(V, np.float32); rgb = np.zeros((V, 3), np.float32)
    for F, d, c in zip(z["full_proj"], z["depth"], z["rgb"]):
        oracle.tsdf_integrate(z["points"], F, d, c, 0.0, tsdf, w, rgb, trunc_pp=z["sdf_trunc"])
    assert np.abs(tsdf - z["tsdf"]).max() < 2e-4 and (tsdf != 1).mean() > 0.2      # float32 bilinear (d - z) cancellation / truncation
    Vv = z["verts"].shape[0]
    tsdf = np.ones(Vv, np.float32); w = np.ones(Vv, np.float32); rgb = np.zeros((Vv, 3), np.float32)
    for F, d, c in zip(z["full_proj"], z["depth"], z["rgb"]):
        oracle.tsdf_integrate(z["verts"], F, d, c, 5 * float(z["voxel_size"]), tsdf, w, rgb)
    assert np.abs(rgb - z["vert_rgb"]).max() < 2e-4 and (rgb != 0).mean() > 0.2


def test_plane_multiview_oracle_matches_reference_run():
    """oracle/gsm_oracle.c vs PGSRScene.get_loss_dict's multi-view branch (geo consistency + patch NCC) on two views of a textured plane."""
    import oracle_multiview as om
    z = golden_ref.load("ref_loss_plane_multiview")
    cfg = om.fixture_cfg(z)
    g = om.geo(cfg, z["plane_depth"], z["near_plane_depth"])
    cnt = g["stats"][1]
    assert 0.5 * cfg.W * cfg.H < cnt < 0.9 * cfg.W * cfg.H                       # the noise threshold and the frustum both cut
    lam = float(z["lambda_geo"])
    np.testing.assert_allclose(lam * g["stats"][0] / cnt, float(z["geo_loss"]), rtol=2e-4)
    idx = np.nonzero(g["dmask"])[0].astype(np.int32)                             # fewer than nunm_sample: the reference takes every valid pixel
    n = om.ncc(cfg, idx, g["weight"], z["rendered_normal"], z["rendered_distance"], z["gray"], z["near_gray"])
    lam_n = float(z["lambda_ncc"])
    assert n["stats"][1] > 0.5 * idx.size
    np.testing.assert_allclose(lam_n * n["stats"][0] / n["stats"][1], float(z["ncc_loss"]), rtol=5e-4)
    H, W = int(z["H"]), int(z["W"])
    for got, exp, name in ((lam / cnt * g["g_depth"], z["d_plane_depth"], "depth"), (lam / cnt * g["g_near"], z["d_near_plane_depth"], "near"),
                           (lam_n / n["stats"][1] * n["g_normal"], z["d_rendered_normal"], "normal"),
                           (lam_n / n["stats"][1] * n["g_dist"], z["d_rendered_distance"], "dist")):
        exp = exp.reshape(got.shape)
        rel = np.linalg.norm(got - exp) / np.linalg.norm(exp)
        assert rel < 2e-3, (name, rel, np.abs(got - exp).max(), np.abs(exp).max())


def test_multiview_torch_restatement_matches_reference_run():
    """tests/ref_mv_torch.py (used for timing and full-size GPU parity) against the same reference-run fixture."""
    import torch
    import ref_mv_torch
    z = golden_ref.load("ref_loss_plane_multiview")
    cam = lambda pre: {k: (z[f"{pre}_{k}"] if k in ("R", "T") else float(z[f"{pre}_{k}"])) for k in ("R", "T", "Fx", "Fy", "Cx", "Cy")}
    lv = {k: torch.tensor(z[k], requires_grad=True) for k in ("plane_depth", "near_plane_depth", "rendered_normal", "rendered_distance")}
    geo, ncc = ref_mv_torch.multiview_loss(lv["plane_depth"], lv["near_plane_depth"], lv["rendered_normal"], lv["rendered_distance"],
                                           torch.tensor(z["gray"]), torch.tensor(z["near_gray"]), cam("v"), cam("n"))
    (geo + ncc).backward()
    np.testing.assert_allclose([geo.item(), ncc.item()], [float(z["geo_loss"]), float(z["ncc_loss"])], rtol=1e-5)
    for k in lv:
        e = z["d_" + k]
        assert np.linalg.norm(lv[k].grad.numpy() - e) / np.linalg.norm(e) < 1e-4, k


def test_plane_allmap_oracle_matches_reference_run():
    """refm_plane_allmap[_bwd] vs PGSRScene.render()'s per-Gaussian all_map (captured from a stub rasterizer) and its autograd."""
    import oracle_multiview as om
    z = golden_ref.load("ref_plane_allmap")
    am, dx, dq = om.plane_allmap(z["means3D"], z["rotations"], z["scales"], z["viewmatrix"], z["campos"], z["dL_dall_map"])
    np.testing.assert_allclose(am, z["all_map"], rtol=2e-5, atol=2e-6)
    assert (am[:, 3] == 1).all()
    assert np.abs(dx - z["d_means3D"]).max() <= 2e-5 * np.abs(z["d_means3D"]).max()
    assert np.abs(dq - z["d_rotations"]).max() <= 5e-5 * np.abs(z["d_rotations"]).max()


def _stats_fixture():
    z = golden_ref.load("ref_training_stats")
    names = ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom")
    calls = [{n: z[f"c{i}_{n}"] for n in ("visible", "neural_opacity", "selection", "update_filter", "grad")} for i in range(2)]
    return z, names, calls


def test_training_stats_oracle_matches_reference_run():
    """refd_training_stats vs ScaffoldGaussian.training_statis run twice on the reference's own model object."""
    z, names, calls = _stats_fixture()
    acc = {n: z["before_" + n].reshape(-1).copy() for n in names}
    for c in calls:
        oracle_decode.training_stats(np.nonzero(c["visible"])[0], int(z["k"]), c["neural_opacity"], c["selection"], c["update_filter"], c["grad"],
                                     *[acc[n] for n in names])
    for n in names:
        np.testing.assert_allclose(acc[n], z["after_" + n].reshape(-1), rtol=1e-6, atol=1e-6, err_msg=n)
        assert not np.array_equal(acc[n], z["before_" + n].reshape(-1))


def test_densify_stats_oracle_matches_reference_run():
    """refm_densify_stats vs VanillaGaussian.densify / PGSRGaussian.densify (statistics-only step) on the reference's own classes."""
    import oracle_multiview as om
    z = golden_ref.load("ref_densify_stats")
    for tag, names in (("vanilla", ("max_radii2D", "xyz_gradient_accum", "denom")),
                       ("pgsr", ("max_radii2D", "xyz_gradient_accum", "denom", "xyz_gradient_accum_abs", "denom_abs"))):
        a = {n: z["init_" + n].reshape(-1).copy() for n in names}
        if tag == "vanilla":
            om.densify_stats(z["visibility_filter"], z["radii"], z["grad"], a["max_radii2D"], a["xyz_gradient_accum"], a["denom"])
        else:
            om.densify_stats(z["visibility_filter"], z["radii"], z["grad"], a["max_radii2D"], a["xyz_gradient_accum"], a["denom"], z["out_observe"],
                             z["grad_abs"], a["xyz_gradient_accum_abs"], a["denom_abs"])
        for n in names:
            np.testing.assert_allclose(a[n], z[f"{tag}_{n}"].reshape(-1), rtol=1e-6, atol=1e-6, err_msg=f"{tag}:{n}")


def _cov_scene():
    import scenes
    z = golden_ref.load("ref_cov3d")
    sc = scenes.make_scene("ewa", z["scales"].shape[0], 160, 120, seed=3, color_mode="precomp", scale_modifier=float(z["scale_modifier"]))
    a = dict(sc); a["scales"] = z["scales"]; a["rotations"] = z["rotations"]; a["cov3D_precomp"] = None
    b = dict(a); b["scales"] = None; b["rotations"] = None; b["cov3D_precomp"] = z["cov3D"]
    return z, a, b


def test_cov3d_oracle_matches_reference_python_twin():
    """The oracle's computeCov3D (scales + rotations + scale_modifier) vs the reference's own Python twin of it
    (build_scaling_rotation / strip_symmetric, general_utils.py:64-110) fed back in as cov3D_precomp: same radii, same image."""
    z, a, b = _cov_scene()
    with oracle.Forward(a, "ewa") as fa, oracle.Forward(b, "ewa") as fb:
        assert np.array_equal(fa.radii, fb.radii) and (fa.radii > 0).sum() > 50
        np.testing.assert_allclose(fa.color, fb.color, rtol=0, atol=2e-5)
    # geom_transform_points pins the row-vector projection convention the preprocess uses (p_hom = [x y z 1] @ full_proj, / w)
    ph = np.concatenate([z["points"], np.ones((z["points"].shape[0], 1), np.float32)], 1) @ z["full_proj_transform"]
    np.testing.assert_allclose(ph[:, :3] / (ph[:, 3:] + 1e-7), z["points_ndc"], rtol=1e-5, atol=1e-6)
