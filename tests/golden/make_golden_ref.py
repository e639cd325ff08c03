"""Golden vectors produced by RUNNING the reference's own Python (torch, CPU) in the authoring container.

    python tests/golden/make_golden_ref.py          # needs /root/reference; writes tests/golden/ref_*.npz

The reference's scene / gaussian classes import CUDA extension modules and a few packages this image lacks (cv2, plyfile, tyro ...);
those imports are satisfied with inert MagicMock stand-ins so that the pure-torch methods below can be called.  Nothing here builds or
replaces the CUDA rasterizers: wherever the reference would call one, the generator hands it a seeded synthetic raster output (the
"allmap"), and what is pinned is the reference's *Python* math around it.  The committed .npz files are data only (inputs + expected
outputs and gradients); no reference source travels.

Pinned (reference file:line -> fixture):
  ref_decode_{scaffold,scaffold_dist,octree}.npz
        gssr/scene/scaffold_scene.py:27-122, gssr/scene/octree_scene.py:26-128  generate_neural_gaussians(is_training=True) on the
        reference's own ScaffoldGaussian / OctreeGaussian MLP stacks (gssr/gaussian/scaffold_gaussian.py:141-159) + autograd of
        sum(out * dL) back to anchors / features / offsets / scaling / every MLP weight / the appearance row.
  ref_lod_{floor,round,ceil,progressive}.npz
        gssr/gaussian/octree_gaussian.py:184-203,255-267 set_anchor_mask + map_to_int_level.
  ref_loss_l1_ssim.npz
        gssr/scene/vanilla_scene.py:29-69 get_loss_dict (l1_loss, ssim) + autograd to the image.
  ref_loss_surfel_geo_r{0,1}.npz
        gssr/scene/twodgs_scene.py:25-35 get_loss_dict(step=8000) over gssr/scene/twodgs_scene.py:83-128 render() post-processing
        (depth_to_normal: gssr/utils/point_utils.py:9-36) with the rasterizer call returning the fixture's allmap; autograd to allmap.
  ref_tsdf_unbounded.npz
        gssr/utils/mesh_utils.py:195-246 compute_sdf_perframe / compute_unbounded_tsdf, driven through extract_mesh_unbounded (:182-277)
        with marching_cubes_with_contraction replaced by a probe that evaluates the sdf callable on the fixture's samples: the
        contracted pass (adaptive truncation) gives `tsdf`, the texturing pass (scalar truncation) gives `rgb`.
  ref_loss_plane_multiview.npz
        gssr/scene/pgsr_scene.py:97-204 get_loss_dict(step=8000, near_cam=..., nearest_render_pkg=...): the multi-view geometric
        consistency loss and the patch NCC loss (get_points_from_depth / get_points_depth_in_depth_map, gssr/utils/point_utils.py:38-75;
        patch_offsets / patch_warp, gssr/utils/graphics_utils.py:185-198; lncc :60-95) on a two-camera view of a textured plane, all
        valid pixels sampled (fewer than nunm_sample, so np.random.choice is not reached); autograd to both plane-depth maps, the
        rendered normal and the rendered distance.
  ref_cov3d.npz
        gssr/utils/general_utils.py:64-110 build_rotation / build_scaling_rotation / strip_symmetric composed as the reference's
        build_covariance_from_scaling_rotation (vanilla_gaussian.py:53-57): the Python twin of the rasterizers' computeCov3D
        (3DGS/cuda_rasterizer/forward.cu:116-147), i.e. the `cov3D_precomp` a caller may pass instead of scales + rotations.  Also
        geom_transform_points (gssr/utils/graphics_utils.py:22-29) on the same cameras as camera_*.npz.
  ref_densify_stats.npz
        gssr/gaussian/vanilla_gaussian.py:467-472,428-430 and gssr/gaussian/pgsr_gaussian.py:164-172,157-161: VanillaGaussian.densify /
        PGSRGaussian.densify called at a step that is not a densification / opacity-reset step (statistics only), accumulators before / after.
  ref_training_stats.npz
        gssr/gaussian/scaffold_gaussian.py:488-508 ScaffoldGaussian.training_statis (the per-iteration densification statistics of the
        Scaffold / Octree methods) run twice on the reference's own model object: accumulators before / after.
  ref_plane_allmap.npz
        gssr/scene/pgsr_scene.py:241-257 (get_smallest_axis / get_normal) and :297-304: the per-Gaussian `all_map` input PGSRScene.render()
        hands to the plane rasterizer, captured from a stub rasterizer, + autograd of sum(all_map * dL) to means3D / rotations.
        quaternion_to_matrix comes from pytorch3d (absent third-party dependency, unpinned in requirements.txt): the generator supplies its
        published formula.
  ref_ply_layout.npz
        gssr/gaussian/vanilla_gaussian.py:140-214, scaffold_gaussian.py:388-456, octree_gaussian.py:276-360: save_gaussians / load_gaussians of
        the three models with plyfile's describe / read replaced by probes (the vertex table's column names and values out, the same table
        back in): the point-cloud column layout and the flatten / reshape conventions of both directions.
  ref_loss_plane_geo.npz
        gssr/scene/pgsr_scene.py:227-238 render_normal (normal_from_depth_image, gssr/utils/graphics_utils.py:139-146),
        pgsr_scene.py:32-58 _get_img_grad_weight / erode, combined exactly as pgsr_scene.py:108-112 (the single-view normal loss);
        autograd to plane_depth and out_all_map.
"""
import importlib
import math
import os
import sys
import types
import warnings
from unittest import mock

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/ : decode_cases, scenes
STUBBED = []


def ref_import(name):
    while True:
        try:
            return importlib.import_module(name)
        except ModuleNotFoundError as e:
            if e.name.startswith("gssr"):
                raise
            STUBBED.append(e.name)
            sys.modules[e.name] = mock.MagicMock()
            for k in [k for k in sys.modules if k.startswith("gssr")]:
                del sys.modules[k]


ref_import("gssr.configs.method_config")           # the entry the reference's own train.py imports first (resolves its import cycle)

# the reference hard-codes device="cuda" / .cuda() in its torch helpers; run them on the CPU
torch.Tensor.cuda = lambda self, *a, **k: self
_to = torch.Tensor.to
torch.Tensor.to = lambda self, *a, **k: _to(self, *[("cpu" if isinstance(x, str) and x.startswith("cuda") else x) for x in a], **k)
_arange = torch.arange
torch.arange = lambda *a, **k: _arange(*a, **{**k, "device": "cpu"}) if "device" in k else _arange(*a, **k)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: v for k, v in arrays.items() if v is not None})
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


# ----------------------------------------------------------------------------------------------------------------- decode
def _fill_linear(lin, W, b):
    with torch.no_grad():
        lin.weight.copy_(torch.tensor(W)); lin.bias.copy_(torch.tensor(b))


def decode_fixture(name, octree, **kw):
    import decode_cases
    case = decode_cases.make_case(Na=160, **kw)
    bank = bool(kw.get("feat_bank"))
    p = case["params"]
    A = 0 if p["app"] is None else p["app"].size
    if octree:
        mod = ref_import("gssr.gaussian.octree_gaussian"); scn = ref_import("gssr.scene.octree_scene")
        cfg = mod.OctreeGaussianConfig(); Gauss, Scene = mod.OctreeGaussian, scn.OctreeScene
        cfg.add_level = case["level"] is not None
        cfg.dist2level = "progressive" if case["opacity_scale"] is not None else "round"
    else:
        mod = ref_import("gssr.gaussian.scaffold_gaussian"); scn = ref_import("gssr.scene.scaffold_scene")
        cfg = mod.ScaffoldGaussianConfig(); Gauss, Scene = mod.ScaffoldGaussian, scn.ScaffoldScene
    cfg.n_offsets = case["k"]; cfg.appearance_dim = A; cfg.use_feat_bank = bank
    cfg.add_opacity_dist, cfg.add_cov_dist, cfg.add_color_dist = case["dist_o"], case["dist_c"], case["dist_k"]
    g = Gauss(cfg, device="cpu")
    uid = 2
    if A:
        g.set_appearance(5)
        emb = [q for q in g.embedding_appearance.parameters()][0]
        with torch.no_grad():
            emb[uid].copy_(torch.tensor(p["app"]))
    _fill_linear(g.mlp_opacity[0], p["W1o"], p["b1o"]); _fill_linear(g.mlp_opacity[2], p["W2o"], p["b2o"])
    _fill_linear(g.mlp_cov[0], p["W1c"], p["b1c"]); _fill_linear(g.mlp_cov[2], p["W2c"], p["b2c"])
    _fill_linear(g.mlp_color[0], p["W1k"], p["b1k"]); _fill_linear(g.mlp_color[2], p["W2k"], p["b2k"])
    if bank:
        _fill_linear(g.mlp_feature_bank[0], p["W1b"], p["b1b"]); _fill_linear(g.mlp_feature_bank[2], p["W2b"], p["b2b"])
    leaf = lambda a: torch.tensor(a).requires_grad_(True)
    g._anchor = leaf(case["anchor"]); g._anchor_feat = leaf(case["feat"]); g._offset = leaf(case["offset"])
    g._scaling = leaf(np.log(case["scaling"]))                     # get_scaling = exp(_scaling)
    Na = case["anchor"].shape[0]
    if octree:
        g._level = torch.tensor(case["level"]).reshape(Na, 1)
        if case["opacity_scale"] is not None:
            r = np.random.default_rng(77)
            trans = r.uniform(size=Na) < 0.5
            prog = np.where(trans, case["opacity_scale"], r.uniform(0.1, 0.9, Na)).astype(np.float32)   # ignored where ~trans
            g._prog_ratio = torch.tensor(prog).reshape(Na, 1); g.transition_mask = torch.tensor(trans)
            case["opacity_scale"] = np.where(trans, prog, 1.0).astype(np.float32)
    vmask = torch.zeros(Na, dtype=torch.bool); vmask[torch.tensor(case["vis_idx"], dtype=torch.long)] = True
    cam = types.SimpleNamespace(camera_center=torch.tensor(case["campos"]), uid=uid)
    fake = types.SimpleNamespace(_gaussians=g)
    xyz, color, opacity, scaling, rot, nop, mask = Scene.generate_neural_gaussians(fake, cam, visible_mask=vmask, is_training=True)
    P = xyz.shape[0]
    dL = decode_cases.make_out_grads(P, seed=kw.get("seed", 0))
    loss = ((xyz * torch.tensor(dL["xyz"])).sum() + (color * torch.tensor(dL["color"])).sum() +
            (opacity.reshape(-1) * torch.tensor(dL["opacity"])).sum() + (scaling * torch.tensor(dL["scaling"])).sum() +
            (rot * torch.tensor(dL["rot"])).sum())
    loss.backward()
    gr = {"g_anchor": g._anchor.grad, "g_feat": g._anchor_feat.grad, "g_offset": g._offset.grad,
          "g_scaling": g._scaling.grad / torch.tensor(case["scaling"])}      # d/d get_scaling = d/d _scaling / exp(_scaling)
    for head, m in (("o", g.mlp_opacity), ("c", g.mlp_cov), ("k", g.mlp_color)):
        gr[f"g_W1{head}"], gr[f"g_b1{head}"], gr[f"g_W2{head}"], gr[f"g_b2{head}"] = m[0].weight.grad, m[0].bias.grad, m[2].weight.grad, m[2].bias.grad
    if bank:
        m = g.mlp_feature_bank
        gr["g_W1b"], gr["g_b1b"], gr["g_W2b"], gr["g_b2b"] = m[0].weight.grad, m[0].bias.grad, m[2].weight.grad, m[2].bias.grad
    if A:
        gr["g_app"] = emb.grad[uid]
        assert not emb.grad[[0, 1, 3, 4]].any()
    margin = float(nop.detach().abs().min())
    assert margin > 1e-4, f"{name}: opacity gate too close to 0 ({margin}); pick another seed"
    flat = {f"in_{n}": case[n] for n in ("anchor", "feat", "offset", "scaling", "level", "opacity_scale", "campos", "vis_idx")}
    flat.update({f"p_{n}": v for n, v in p.items()})
    flat.update({f"dL_{n}": v for n, v in dL.items()})
    flat.update({n: v.detach().numpy() for n, v in gr.items()})
    save(name, k=case["k"], dist_o=case["dist_o"], dist_c=case["dist_c"], dist_k=case["dist_k"], gate_margin=margin,
         xyz=xyz.detach().numpy(), color=color.detach().numpy(), opacity=opacity.detach().numpy(), scaling=scaling.detach().numpy(),
         rot=rot.detach().numpy(), neural_opacity=nop.detach().numpy().reshape(-1), mask=mask.numpy(), **flat)


def lod_fixture(mode):
    mod = ref_import("gssr.gaussian.octree_gaussian")
    cfg = mod.OctreeGaussianConfig(); cfg.dist2level = mode; cfg.progressive = False
    g = mod.OctreeGaussian(cfg, device="cpu")
    r = np.random.default_rng(3)
    Na, levels, fork, vs, sd, rs = 3000, 6, 2, 0.4, 12.0, 1.0
    anchor = r.uniform(-8, 8, (Na, 3)).astype(np.float32); level = r.integers(0, levels, Na).astype(np.int32)
    extra = r.uniform(-0.3, 0.3, Na).astype(np.float32); campos = np.array([0.5, -1.0, 3.0], np.float32)
    g._anchor = torch.tensor(anchor); g._level = torch.tensor(level).unsqueeze(1); g._extra_level = torch.tensor(extra)
    g.voxel_size, g.fork, g.standard_dist, g.levels = vs, fork, sd, levels
    g.set_anchor_mask(torch.tensor(campos), 0, rs)
    out = {"anchor_mask": g._anchor_mask.numpy()}
    if mode == "progressive":
        out["prog_ratio"] = g._prog_ratio.numpy().reshape(-1); out["transition_mask"] = g.transition_mask.numpy()
    save(f"ref_lod_{mode}.npz", anchor=anchor, level=level, extra_level=extra, campos=campos, voxel_size=vs, fork=fork, standard_dist=sd,
         resolution_scale=rs, levels=levels, **out)


# ----------------------------------------------------------------------------------------------------------------- losses
def _bare(cls, **attrs):
    o = object.__new__(cls)
    for k, v in attrs.items():
        object.__setattr__(o, k, v)
    return o


def l1_ssim_fixture():
    scn = ref_import("gssr.scene.vanilla_scene")
    r = np.random.default_rng(21)
    shape = (3, 45, 61)
    gt = r.uniform(0, 1, shape).astype(np.float32)
    img = np.clip(gt + r.normal(0, 0.12, shape), 0, 1).astype(np.float32)
    scene = _bare(scn.VanillaScene, config=types.SimpleNamespace(lambda_dssim=0.2), device="cpu")
    x = torch.tensor(img, requires_grad=True)
    cam = types.SimpleNamespace(original_image=torch.tensor(gt))
    d = scene.get_loss_dict({"render": x}, cam, step=1)
    total = d["L1_loss"] + d["ssim_loss"]
    total.backward()
    save("ref_loss_l1_ssim.npz", image=img, gt=gt, lambda_dssim=0.2, L1_loss=d["L1_loss"].item(), ssim_loss=d["ssim_loss"].item(),
         total=total.item(), d_image=x.grad.numpy())


def _camera(W, H):
    import scenes
    cam = scenes.make_camera(W, H, 0.8 * W, 0.8 * W, yaw_deg=20.0, t=(0.5, 0.2, 0.0))
    wvt = torch.tensor(cam["viewmatrix"]); fpt = torch.tensor(cam["projmatrix"])
    return cam, types.SimpleNamespace(world_view_transform=wvt, full_proj_transform=fpt, image_width=W, image_height=H,
                                      FoVx=2 * math.atan(cam["tanfovx"]), FoVy=2 * math.atan(cam["tanfovy"]),
                                      camera_center=wvt.inverse()[3, :3])


def _allmap(H, W, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = 3.0 + 0.5 * np.sin(xx / 7.0) + 0.3 * np.cos(yy / 5.0) + r.normal(0, 0.02, (H, W))
    alpha = r.uniform(0.3, 1.0, (H, W))
    alpha[r.uniform(size=(H, W)) < 0.04] = 0.0                     # empty pixels: the reference's 0/0 -> nan_to_num path
    am = np.zeros((7, H, W), np.float32)
    n = r.normal(0, 1, (3, H, W)); n /= np.linalg.norm(n, axis=0, keepdims=True)
    am[0] = depth * alpha; am[1] = alpha; am[2:5] = n * alpha; am[5] = depth + r.normal(0, 0.05, (H, W)); am[6] = r.uniform(0, 0.1, (H, W))
    return am.astype(np.float32)


def surfel_geo_fixture(depth_ratio):
    scn = ref_import("gssr.scene.twodgs_scene")
    H, W = 37, 52
    am = _allmap(H, W, 5 + int(depth_ratio))
    camd, cam = _camera(W, H)
    x = torch.tensor(am, requires_grad=True)
    img = torch.zeros(3, H, W)
    # the one non-Python step of render(): the rasterizer call returns the fixture's raster output
    scn.GaussianRasterizationSettings = lambda **kw: None
    scn.GaussianRasterizer = lambda raster_settings: (lambda **kw: (img, torch.ones(4), x))
    lam_n, lam_d = 0.05, 100.0
    scene = _bare(scn.TwoDGSScene, device="cpu", background=torch.zeros(3),
                  config=types.SimpleNamespace(lambda_dssim=0.2, lambda_normal=lam_n, lambda_dist=lam_d, depth_ratio=depth_ratio,
                                               scaling_modifier=1.0, debug=False),
                  _gaussians=types.SimpleNamespace(active_sh_degree=0))
    out = scene.render(cam, torch.zeros(4, 3), None, None, None, None, None, None)
    cam.original_image = torch.zeros(3, H, W)
    d = scene.get_loss_dict(out, cam, step=8000)
    total = d["normal_loss"] + d["dist_loss"]
    total.backward()
    g = x.grad.numpy()
    bad = ~np.isfinite(g)                                          # autograd of x/0 -> nan_to_num: NaN on ch 0/1 at alpha == 0 only
    assert not bad[2:].any() and (am[1] == 0)[bad[0] | bad[1]].all()
    save(f"ref_loss_surfel_geo_r{int(depth_ratio)}.npz", allmap=am, viewmatrix=camd["viewmatrix"], projmatrix=camd["projmatrix"], W=W, H=H,
         depth_ratio=depth_ratio, lambda_normal=lam_n, lambda_dist=lam_d, normal_loss=d["normal_loss"].item(), dist_loss=d["dist_loss"].item(),
         surf_depth=out["depth"].detach().numpy(), surf_normal=out["surf_normal"].detach().numpy(), normal=out["normal"].detach().numpy(),
         d_allmap=np.where(bad, 0.0, g).astype(np.float32), d_allmap_nan=bad)


def plane_geo_fixture():
    scn = ref_import("gssr.scene.pgsr_scene")
    H, W = 41, 47
    r = np.random.default_rng(9)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = (3.0 + 0.5 * np.sin(xx / 7.0) + 0.3 * np.cos(yy / 5.0) + r.normal(0, 0.02, (H, W))).astype(np.float32)
    alpha = r.uniform(0.2, 1.0, (H, W)); n = r.normal(0, 1, (3, H, W)); n /= np.linalg.norm(n, axis=0, keepdims=True)
    oam = np.zeros((5, H, W), np.float32); oam[0:3] = n * alpha; oam[3] = alpha; oam[4] = r.uniform(0, 1, (H, W))
    gt = r.uniform(0, 1, (3, H, W)).astype(np.float32)
    gt[:, :, : W // 2] = gt[:, :, : W // 2] * 0.05 + 0.4          # a flat half: non-trivial image-gradient weights after erosion
    fx = fy = 0.8 * W; cx, cy = W / 2.0, H / 2.0
    camd, cam = _camera(W, H)
    cam.Fx, cam.Fy, cam.Cx, cam.Cy = fx, fy, cx, cy
    Camera = ref_import("gssr.cameras").Camera
    cam.get_calib_matrix_nerf = types.MethodType(Camera.get_calib_matrix_nerf, cam)
    scene = _bare(scn.PGSRScene, device="cpu", config=types.SimpleNamespace(lambda_normal=0.015))
    pd = torch.tensor(depth[None], requires_grad=True); om = torch.tensor(oam, requires_grad=True)
    # pgsr_scene.py:316-320
    rendered_normal = om[0:3]; rendered_alpha = om[3:4]
    depth_normal = scene.render_normal(cam, pd.squeeze()) * rendered_alpha.detach()
    # pgsr_scene.py:108-112
    image_weight = 1.0 - scene._get_img_grad_weight(torch.tensor(gt))
    image_weight = image_weight.clamp(0, 1).detach() ** 5
    image_weight = scene.erode(image_weight[None, None]).squeeze()
    normal_loss = scene.config.lambda_normal * (image_weight * ((depth_normal - rendered_normal).abs().sum(0))).mean()
    normal_loss.backward()
    save("ref_loss_plane_geo.npz", plane_depth=depth[None], out_all_map=oam, gt_image=gt, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H,
         viewmatrix=camd["viewmatrix"], projmatrix=camd["projmatrix"], lambda_normal=0.015, image_weight=image_weight.numpy(),
         depth_normal=depth_normal.detach().numpy(), normal_loss=normal_loss.item(), d_plane_depth=pd.grad.numpy(), d_out_all_map=om.grad.numpy())


def _plane_view(Camera, W, H, yaw, t, n_w, c_w, phase):
    """A reference-Camera-shaped object looking at the world plane n_w . X = c_w, with its analytic depth / normal / distance / texture."""
    import scenes
    cam = scenes.make_camera(W, H, 0.9 * W, 0.9 * W, yaw_deg=yaw, t=t)
    wvt = cam["viewmatrix"].astype(np.float64)
    R, T = wvt[:3, :3].copy(), wvt[3, :3].copy()                   # cameras/__init__.py:85 with trans = 0, scale = 1
    o = types.SimpleNamespace(R=R, T=T, world_view_transform=torch.tensor(wvt, dtype=torch.float32), image_width=W, image_height=H,
                              Fx=0.9 * W, Fy=0.9 * W, Cx=0.5 * W, Cy=0.5 * H, ncc_scale=1.0)
    for m in ("get_rays", "get_k", "get_inv_k"):
        setattr(o, m, types.MethodType(getattr(Camera, m), o))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    ray = np.stack([(xx - o.Cx) / o.Fx, (yy - o.Cy) / o.Fy, np.ones_like(xx)], -1)
    n_c = n_w @ R; c_c = c_w + n_c @ T
    d0 = c_c / (ray @ n_c)
    Xw = (ray * d0[..., None] - T) @ R.T
    a, b = Xw[..., 0], Xw[..., 1]
    gray = 0.5 + 0.2 * np.sin(5 * a) + 0.2 * np.cos(7 * b) + 0.1 * np.sin(11 * (a + b))
    depth = d0 * (1 + 0.12 * np.sin(xx / 5.0 + phase) * np.cos(yy / 4.0))
    sgn = -1.0 if (ray[H // 2, W // 2] @ n_c) > 0 else 1.0      # rendered normals face the camera
    return o, depth.astype(np.float32), (sgn * n_c).astype(np.float32), float(abs(c_c)), gray.astype(np.float32)


def plane_multiview_fixture():
    scn = ref_import("gssr.scene.pgsr_scene")
    Camera = ref_import("gssr.cameras").Camera
    r = np.random.default_rng(41)
    W, H = 72, 54
    n_w = np.array([0.12, -0.2, -1.0]); n_w /= np.linalg.norm(n_w); c_w = float(n_w @ np.array([0.0, 0.0, 3.0]))
    vc, dv, nv, distv, gv = _plane_view(Camera, W, H, 4.0, (0.05, 0.0, 0.0), n_w, c_w, 0.0)
    nc, dn, _, _, gn = _plane_view(Camera, W, H, -5.0, (-0.25, 0.04, 0.03), n_w, c_w, 1.3)
    normal = (nv[:, None, None] + r.normal(0, 0.01, (3, H, W))).astype(np.float32)
    dist = (distv * (1 + r.normal(0, 0.002, (H, W)))).astype(np.float32)
    gt = np.repeat(gv[None], 3, 0)
    vc.original_image = torch.tensor(gt); vc.gray_image = torch.tensor(gv[None])
    nc.gray_image = torch.tensor(gn[None])
    cfg = scn.PGSRSceneConfig()
    scene = _bare(scn.PGSRScene, device="cpu", config=types.SimpleNamespace(
        lambda_dssim=0.2, lambda_normal=cfg.lambda_normal, lambda_ncc=cfg.lambda_ncc, lambda_geo=cfg.lambda_geo, patch_size=cfg.patch_size,
        nunm_sample=cfg.nunm_sample, pixel_noise_threshold=cfg.pixel_noise_threshold))
    pd = torch.tensor(dv[None], requires_grad=True); npd = torch.tensor(dn[None], requires_grad=True)
    nm = torch.tensor(normal, requires_grad=True); ds = torch.tensor(dist[None], requires_grad=True)
    outputs = {"render": torch.tensor(gt), "rendered_normal": nm, "depth_normal": nm.detach(), "plane_depth": pd, "rendered_distance": ds}
    d = scene.get_loss_dict(outputs, vc, step=8000, near_cam=nc, nearest_render_pkg={"plane_depth": npd})
    assert float(d["normal_loss"]) == 0.0 and float(d["geo_loss"]) > 0 and float(d["ncc_loss"]) > 0
    (d["geo_loss"] + d["ncc_loss"]).backward()
    camd = lambda o: dict(R=o.R, T=o.T, Fx=o.Fx, Fy=o.Fy, Cx=o.Cx, Cy=o.Cy)
    save("ref_loss_plane_multiview.npz", W=W, H=H, plane_depth=dv[None], near_plane_depth=dn[None], rendered_normal=normal, rendered_distance=dist[None],
         gray=gv[None], near_gray=gn[None], **{f"v_{k}": v for k, v in camd(vc).items()}, **{f"n_{k}": v for k, v in camd(nc).items()},
         lambda_geo=cfg.lambda_geo, lambda_ncc=cfg.lambda_ncc, patch_size=cfg.patch_size, pixel_noise_threshold=cfg.pixel_noise_threshold,
         geo_loss=d["geo_loss"].item(), ncc_loss=d["ncc_loss"].item(), d_plane_depth=pd.grad.numpy(), d_near_plane_depth=npd.grad.numpy(),
         d_rendered_normal=nm.grad.numpy(), d_rendered_distance=ds.grad.numpy())


def cov3d_fixture():
    gu = ref_import("gssr.utils.general_utils"); gr = ref_import("gssr.utils.graphics_utils")
    _z = torch.zeros
    torch.zeros = lambda *a, **k: _z(*a, **{kk: ("cpu" if kk == "device" else v) for kk, v in k.items()})
    r = np.random.default_rng(81)
    P = 256
    scales = np.exp(r.normal(-2.5, 0.8, (P, 3))).astype(np.float32)
    q = r.normal(0, 1, (P, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    mod = 1.3
    L = gu.build_scaling_rotation(mod * torch.tensor(scales), torch.tensor(q))
    cov6 = gu.strip_symmetric(L @ L.transpose(1, 2))
    pts = r.uniform(-3, 3, (P, 3)).astype(np.float32)
    cam = np.load(os.path.join(HERE, "camera_1.npz"))
    proj = gr.geom_transform_points(torch.tensor(pts), torch.tensor(cam["full_proj_transform"]))
    torch.zeros = _z
    save("ref_cov3d.npz", scales=scales, rotations=q, scale_modifier=mod, cov3D=cov6.numpy(), points=pts,
         full_proj_transform=cam["full_proj_transform"], points_ndc=proj.numpy())


def densify_stats_fixture():
    van = ref_import("gssr.gaussian.vanilla_gaussian"); pg = ref_import("gssr.gaussian.pgsr_gaussian")
    r = np.random.default_rng(71)
    P = 900
    cfg = types.SimpleNamespace(densify_until_iter=15000, densify_from_iter=500, densification_interval=100, opacity_reset_interval=3000)
    filt = r.uniform(size=P) < 0.6
    radii = np.where(filt, r.integers(1, 60, P), 0).astype(np.int32)
    obs = r.integers(0, 3, P).astype(np.int32)
    grad = r.normal(0, 1, (P, 3)).astype(np.float32); grad_abs = np.abs(r.normal(0, 1, (P, 3))).astype(np.float32)
    init = {"max_radii2D": r.uniform(0, 40, P).astype(np.float32), "xyz_gradient_accum": r.uniform(0, 1, (P, 1)).astype(np.float32),
            "denom": r.integers(0, 5, (P, 1)).astype(np.float32), "xyz_gradient_accum_abs": r.uniform(0, 1, (P, 1)).astype(np.float32),
            "denom_abs": r.integers(0, 5, (P, 1)).astype(np.float32)}
    out = {}
    for tag, cls, names in (("vanilla", van.VanillaGaussian, ("max_radii2D", "xyz_gradient_accum", "denom")),
                            ("pgsr", pg.PGSRGaussian, tuple(init))):
        g = _bare(cls, config=cfg)
        for n in names:
            object.__setattr__(g, n, torch.tensor(init[n]))
        kw = dict(visibility_filter=torch.tensor(filt), radii=torch.tensor(radii), viewspace_points=types.SimpleNamespace(grad=torch.tensor(grad)))
        if tag == "pgsr":
            kw.update(out_observe=torch.tensor(obs), viewspace_points_abs=types.SimpleNamespace(grad=torch.tensor(grad_abs)))
        g.densify(101, **kw)
        out.update({f"{tag}_{n}": getattr(g, n).numpy() for n in names})
    save("ref_densify_stats.npz", visibility_filter=filt, radii=radii, out_observe=obs, grad=grad, grad_abs=grad_abs,
         **{f"init_{n}": v for n, v in init.items()}, **out)


def training_stats_fixture():
    mod = ref_import("gssr.gaussian.scaffold_gaussian")
    cfg = mod.ScaffoldGaussianConfig(); cfg.n_offsets = 6
    g = mod.ScaffoldGaussian(cfg, device="cpu")
    r = np.random.default_rng(61)
    Na, k = 400, 6
    g.opacity_accum = torch.tensor(r.uniform(0, 2, (Na, 1)).astype(np.float32))
    g.anchor_demon = torch.tensor(r.integers(0, 5, (Na, 1)).astype(np.float32))
    g.offset_gradient_accum = torch.tensor(r.uniform(0, 1, (Na * k, 1)).astype(np.float32))
    g.offset_denom = torch.tensor(r.integers(0, 4, (Na * k, 1)).astype(np.float32))
    before = {n: getattr(g, n).numpy().copy() for n in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom")}
    calls = []
    for it in range(2):
        vis = r.uniform(size=Na) < 0.6
        Nv = int(vis.sum())
        nop = np.tanh(r.normal(0, 1, (Nv * k, 1))).astype(np.float32)
        sel = (nop > 0).reshape(-1)
        P = int(sel.sum())
        upd = r.uniform(size=P) < 0.7
        grad = r.normal(0, 1, (P, 3)).astype(np.float32)
        vsp = types.SimpleNamespace(grad=torch.tensor(grad))
        g.training_statis(vsp, torch.tensor(nop), torch.tensor(upd), torch.tensor(sel), torch.tensor(vis))
        calls.append(dict(visible=vis, neural_opacity=nop.reshape(-1), selection=sel, update_filter=upd, grad=grad))
    after = {n: getattr(g, n).numpy().copy() for n in before}
    flat = {}
    for i, c in enumerate(calls):
        flat.update({f"c{i}_{n}": v for n, v in c.items()})
    save("ref_training_stats.npz", Na=Na, k=k, **{f"before_{n}": v for n, v in before.items()}, **{f"after_{n}": v for n, v in after.items()}, **flat)


def _pytorch3d_quaternion_to_matrix(quaternions):
    """pytorch3d.transforms.quaternion_to_matrix (real part first), restated from its published definition."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def plane_allmap_fixture():
    scn = ref_import("gssr.scene.pgsr_scene")
    Camera = ref_import("gssr.cameras").Camera
    scn.quaternion_to_matrix = _pytorch3d_quaternion_to_matrix
    r = np.random.default_rng(51)
    P, W, H = 700, 64, 48
    camd, cam = _camera(W, H)
    cam.Fx, cam.Fy, cam.Cx, cam.Cy = 0.8 * W, 0.8 * W, W / 2.0, H / 2.0
    cam.get_calib_matrix_nerf = types.MethodType(Camera.get_calib_matrix_nerf, cam)
    xyz = np.concatenate([r.uniform(-3, 3, (P, 2)), r.uniform(-2, 8, (P, 1))], 1).astype(np.float32)     # in front of and behind the camera
    q = r.normal(0, 1, (P, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[:50] *= r.uniform(0.5, 2.0, (50, 1)).astype(np.float32)          # un-normalised rows exercise the two_s term
    sc = np.exp(r.normal(-2, 0.7, (P, 3))).astype(np.float32)
    sc[60:70, 1] = sc[60:70, 0]                                          # ties: first minimum wins
    got = {}

    def rasterizer_stub(raster_settings):
        def call(**kw):
            got["all_map"] = kw["all_map"]
            return (torch.zeros(3, H, W), torch.ones(P), torch.zeros(P), torch.ones(5, H, W), torch.ones(1, H, W) * 3)
        return call
    scn.PlaneGaussianRasterizationSettings = lambda **kw: None
    scn.PlaneGaussianRasterizer = rasterizer_stub
    scene = _bare(scn.PGSRScene, device="cpu", background=torch.zeros(3), config=types.SimpleNamespace(scaling_modifier=1.0, debug=False),
                  _gaussians=types.SimpleNamespace(active_sh_degree=0))
    _zl = torch.zeros_like
    torch.zeros_like = lambda t, **kw: _zl(t, **{k: ("cpu" if k == "device" else v) for k, v in kw.items()})
    m = torch.tensor(xyz, requires_grad=True); rq = torch.tensor(q, requires_grad=True); ss = torch.tensor(sc, requires_grad=True)
    scene.render(cam, m, torch.ones(P, 1), ss, rq, None, None, torch.zeros(P, 3))
    torch.zeros_like = _zl
    am = got["all_map"]
    dL = r.normal(0, 1, (P, 5)).astype(np.float32)
    (am * torch.tensor(dL)).sum().backward()
    assert ss.grad is None or not ss.grad.any()
    save("ref_plane_allmap.npz", means3D=xyz, rotations=q, scales=sc, viewmatrix=camd["viewmatrix"], campos=cam.camera_center.numpy(),
         all_map=am.detach().numpy(), dL_dall_map=dL, d_means3D=m.grad.numpy(), d_rotations=rq.grad.numpy())


def tsdf_fixture():
    import scenes
    mu = ref_import("gssr.utils.mesh_utils")
    r = np.random.default_rng(31)
    W, H, nf = 64, 48, 3
    stack, depths, rgbs = [], [], []
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for i in range(nf):
        cam = scenes.make_camera(W, H, 0.9 * W, 0.9 * W, yaw_deg=-15.0 + 15.0 * i, t=(0.2 * i - 0.2, 0.05 * i, 0.1 * i))
        stack.append(types.SimpleNamespace(full_proj_transform=torch.tensor(cam["projmatrix"])))
        depths.append((2.5 + 0.4 * np.sin(xx / 9.0 + i) + 0.3 * np.cos(yy / 6.0) + r.normal(0, 0.01, (H, W))).astype(np.float32)[None])
        rgbs.append(r.uniform(0, 1, (3, H, W)).astype(np.float32))
    center = np.array([0.1, -0.05, 2.6], np.float32); radius = 1.7; N = 64
    ex = _bare(mu.GaussianExtractor, viewpoint_stack=stack, depthmaps=[torch.tensor(d) for d in depths], rgbmaps=[torch.tensor(c) for c in rgbs],
               radius=radius, center=torch.tensor(center), gaussians=types.SimpleNamespace(get_xyz=torch.tensor(r.normal(0, 1, (50, 3)).astype(np.float32)) + torch.tensor(center)))
    samples = r.uniform(-1.6, 1.6, (6000, 3)).astype(np.float32)          # contracted coordinates, inside and outside the unit ball
    samples = samples[np.linalg.norm(samples, axis=1) < 1.95][:2500]
    near = np.stack([r.uniform(-0.8, 0.8, 3500), r.uniform(-0.6, 0.6, 3500), r.uniform(-0.35, 0.25, 3500)], 1)   # around the depth surfaces
    samples = np.concatenate([samples, near]).astype(np.float32)
    verts = (center + r.uniform(-1.2, 1.2, (3000, 3)) * np.array([1.0, 0.8, 0.25])).astype(np.float32)   # world points for the texturing pass
    got = {}

    def probe(sdf, bounding_box_min, bounding_box_max, level, resolution, inv_contraction):
        got["tsdf"] = sdf(torch.tensor(samples)).numpy()
        got["points"] = inv_contraction(torch.tensor(samples)).numpy()
        return types.SimpleNamespace(as_open3d=types.SimpleNamespace(vertices=verts))
    mu.marching_cubes_with_contraction = probe
    mu.tqdm = lambda it, **kw: it
    mu.o3d.utility.Vector3dVector = lambda a: a
    mesh = ex.extract_mesh_unbounded(resolution=N)
    voxel = radius * 2 / N
    nrm = np.linalg.norm(samples, axis=1)
    trunc = (5 * voxel * np.where(nrm > 1, 1.0 / (2.0 - np.minimum(nrm, 1.9)), 1.0)).astype(np.float32)   # mesh_utils.py:213-216 (an INPUT of our op)
    save("ref_tsdf_unbounded.npz", W=W, H=H, full_proj=np.stack([c.full_proj_transform.numpy() for c in stack]), depth=np.stack(depths),
         rgb=np.stack(rgbs), points=got["points"], sdf_trunc=trunc, tsdf=got["tsdf"], verts=verts, voxel_size=voxel, vert_rgb=np.asarray(mesh.vertex_colors))


# ----------------------------------------------------------------------------------------------------------------- point-cloud files
class _PlyVertex:
    """What `plydata.elements[0]` has to offer the reference's load_gaussians: column access by name and `.properties[i].name`."""
    def __init__(self, rec):
        self.rec = rec
        self.properties = [types.SimpleNamespace(name=n) for n in rec.dtype.names]

    def __getitem__(self, name):
        return self.rec[name]


def ply_fixture():
    """VanillaGaussian / ScaffoldGaussian / OctreeGaussian save_gaussians + load_gaussians.  plyfile is absent: `PlyElement.describe` is
    replaced by a probe that captures the structured array the reference hands to it (= the file's vertex table, names and values), and
    `PlyData.read` by one that serves that table back, so what is pinned is the reference's column layout and its (N, C, S) <-> flat
    conventions in both directions -- not plyfile's container, which tests/test_ply_cpu.py checks against the PLY specification."""
    g = torch.Generator().manual_seed(21)
    rnd = lambda *s: torch.randn(*s, generator=g)
    _tensor = torch.tensor
    torch.tensor = lambda *a, **k: _tensor(*a, **{**k, "device": "cpu"}) if "device" in k else _tensor(*a, **k)
    _ones = torch.ones
    torch.ones = lambda *a, **k: _ones(*a, **{**k, "device": "cpu"}) if "device" in k else _ones(*a, **k)
    out = {}
    try:
        for tag, modname, clsname, attrs in (
                ("vanilla", "gssr.gaussian.vanilla_gaussian", "VanillaGaussian",
                 dict(_xyz=rnd(6, 3), _features_dc=rnd(6, 1, 3), _features_rest=rnd(6, 15, 3), _opacity=rnd(6, 1), _scaling=rnd(6, 3), _rotation=rnd(6, 4),
                      max_sh_degree=3)),
                ("scaffold", "gssr.gaussian.scaffold_gaussian", "ScaffoldGaussian",
                 dict(_anchor=rnd(5, 3), _offset=rnd(5, 10, 3), _anchor_feat=rnd(5, 32), _opacity=rnd(5, 1), _scaling=rnd(5, 6), _rotation=rnd(5, 4))),
                ("octree", "gssr.gaussian.octree_gaussian", "OctreeGaussian",
                 dict(_anchor=rnd(5, 3), _offset=rnd(5, 10, 3), _anchor_feat=rnd(5, 32), _opacity=rnd(5, 1), _scaling=rnd(5, 6), _rotation=rnd(5, 4),
                      _level=torch.randint(0, 4, (5, 1), generator=g), _extra_level=torch.rand(5, generator=g), voxel_size=0.03125, standard_dist=6.5))):
            mod = ref_import(modname)
            cls = getattr(mod, clsname)
            obj = _bare(cls, **attrs)
            got = {}
            mod.PlyElement.describe = lambda elements, name: got.update(rec=elements.copy(), element=name)
            obj.save_gaussians("unused.ply")
            rec = got["rec"]
            assert got["element"] == "vertex" and all(rec.dtype[n] == np.dtype("f4") for n in rec.dtype.names)
            out[tag + "_names"] = np.array(rec.dtype.names)
            out[tag + "_table"] = np.stack([rec[n] for n in rec.dtype.names], axis=1).astype(np.float32)
            for k, v in attrs.items():
                if torch.is_tensor(v):
                    out[f"{tag}_in{k}"] = v.numpy()
            mod.PlyData.read = lambda path: types.SimpleNamespace(elements=[_PlyVertex(rec)])
            fresh = _bare(cls, **({"max_sh_degree": 3} if tag == "vanilla" else {}))
            fresh.load_gaussians("unused.ply")
            for k in attrs:
                v = getattr(fresh, k, None)
                if torch.is_tensor(v):
                    out[f"{tag}_loaded{k}"] = v.detach().numpy()
                elif k in ("voxel_size", "standard_dist"):
                    out[f"{tag}_loaded_{k}"] = np.float32(float(v))
    finally:
        torch.tensor = _tensor; torch.ones = _ones
    save("ref_ply_layout.npz", **out)


if __name__ == "__main__":
    ply_fixture()
    cov3d_fixture()
    densify_stats_fixture()
    training_stats_fixture()
    plane_allmap_fixture()
    plane_multiview_fixture()
    tsdf_fixture()
    torch.manual_seed(0)
    decode_fixture("ref_decode_scaffold.npz", octree=False)
    decode_fixture("ref_decode_scaffold_dist.npz", octree=False, A=0, k=5, dist_o=True, dist_c=True, dist_k=True, seed=1)
    decode_fixture("ref_decode_octree.npz", octree=True, level=True, progressive=True, A=16, k=12, dist_k=True, seed=2)
    decode_fixture("ref_decode_scaffold_featbank.npz", octree=False, feat_bank=True, A=8, k=6, seed=4)      # use_feat_bank=True branch
    for m in ("floor", "round", "ceil", "progressive"):
        lod_fixture(m)
    l1_ssim_fixture()
    surfel_geo_fixture(0.0)
    surfel_geo_fixture(1.0)
    plane_geo_fixture()
    print("stubbed imports:", sorted(set(STUBBED)))
