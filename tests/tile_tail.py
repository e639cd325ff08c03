"""BASELINE config 5's tail (TEST INFRASTRUCTURE + tools/bench_tile_tail.py): per scene tile, render every camera inside the tile's box with the
surfel rasterizer, form the surface depth, integrate it into a block-sparse TSDF volume, merge the tiles' volumes.

Follows /root/reference/extract_mesh_split.py:54-119 (tile loop: cameras inside box.txt -> GaussianExtractor.reconstruction -> one ScalableTSDFVolume
over all tiles' frames), gssr/utils/mesh_utils.py:50-75 (to_cam_open3d: fx = W / (2 tan), cx = (W - 1) / 2, extrinsic = world_view_transform^T),
:108-121 (render -> rgb, depth), :154-178 (depth zeroed where the alpha mask is low, colours to uint8, integrate) and
gssr/scene/twodgs_scene.py:96-111 (surface depth = expected depth allmap[0] / alpha, or the median depth allmap[5], by depth_ratio).
numpy / torch only; the HIP side keeps every image on the device between the rasterizer and the volume."""
import math

import numpy as np

import scenes


def lay_on_terrain(sc, seed=0):
    """Moves a SURVEY 8d surfel scene onto ONE smooth surface: depth 4 + 0.6 sin(3 u) + 0.4 cos(2.5 v) (u, v = the surfel's position in the frustum, -1.1 .. 1.1)
    with 2 mm of jitter, discs facing the camera up to a ~15 degree tilt.  The scene of SURVEY 8d is a cloud of independent surfels at depths 1 .. 20: its
    rendered depth map is NOISE (neighbouring pixels blend different surfels at unrelated depths), the worst case for a TSDF volume -- the voxels a frame updates
    are spatially independent.  A trained 2DGS scene renders surfaces; this is the stand-in for it."""
    rng = np.random.default_rng(seed + 991)
    V = sc["viewmatrix"].astype(np.float64)
    pc = sc["means3D"].astype(np.float64) @ V[:3, :3] + V[3, :3]
    u, v = pc[:, 0] / (pc[:, 2] * sc["tanfovx"]), pc[:, 1] / (pc[:, 2] * sc["tanfovy"])
    z = 4.0 + 0.6 * np.sin(3.0 * u) + 0.4 * np.cos(2.5 * v) + rng.normal(0, 0.002, len(u))
    scale = z / pc[:, 2]
    pc = np.stack([u * z * sc["tanfovx"], v * z * sc["tanfovy"], z], -1)
    sc["means3D"] = ((pc - V[3, :3]) @ np.linalg.inv(V[:3, :3])).astype(np.float32)
    sc["scales"] = (sc["scales"] * scale[:, None]).astype(np.float32)          # the same footprint in pixels
    q = np.concatenate([np.ones((len(u), 1)), rng.normal(0, 0.13, (len(u), 3))], 1)
    sc["rotations"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    return sc


def make_tiles(n_tiles=2, cams_per_tile=3, P=6000, W=320, H=208, seed=0, sigma_px=5.0, surface="cloud"):
    """-> list of tiles: dict(gauss = the tile's surfels (rasterizer kwargs without camera), cams = [camera dicts]).  Tile t's surfels sit in the
    frustum of its first camera; its other cameras look at the same surfels from nearby poses.  Tiles are 8 world units apart along x.
    surface: "cloud" = the SURVEY 8d distribution (independent surfels, depth maps of noise), "terrain" = the same surfels laid on one smooth surface."""
    tiles = []
    fx = W * (1600.0 / 1920.0)
    for t in range(n_tiles):
        sc = scenes.make_scene("surfel", P, W, H, fx=fx, seed=seed + 17 * t, sigma_px=sigma_px, bg=(0.0, 0.0, 0.0))
        if surface == "terrain":
            lay_on_terrain(sc, seed + 17 * t)
        off = np.array([8.0 * t, 0.0, 0.0], np.float32)
        gauss = {k: sc[k] for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp", "bg", "scale_modifier", "sh_degree")}
        gauss["means3D"] = (sc["means3D"] + off).astype(np.float32)
        gauss["opacities"] = np.clip(sc["opacities"] * 1.5, 0.0, 0.98).astype(np.float32)       # mostly opaque surfaces: depth maps with few holes
        cams = []
        for k in range(cams_per_tile):
            yaw = 4.0 * (k - (cams_per_tile - 1) / 2)
            a = math.radians(yaw)
            R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float64)
            C = np.array([8.0 * t + 0.25 * k, 0.05 * k, -0.1 * k], np.float64)                  # camera centre in the world
            cams.append(scenes.make_camera(W, H, fx, fx, yaw_deg=yaw, t=tuple((-R.T @ C).tolist())))    # world2view(R, t): p_cam = R^T p + t
        tiles.append(dict(gauss=gauss, cams=cams))
    return tiles


def frame_scene(tile, k):
    sc = dict(tile["cams"][k])
    sc.update(tile["gauss"])
    sc.update(variant="surfel", render_geo=True)
    return sc


def o3d_camera(cam):
    """mesh_utils.py:50-72 -> (fx, fy, cx, cy, extrinsic 4x4 world->camera)."""
    W, H = cam["W"], cam["H"]
    return W / (2.0 * cam["tanfovx"]), H / (2.0 * cam["tanfovy"]), (W - 1) / 2.0, (H - 1) / 2.0, cam["viewmatrix"].astype(np.float64).T.copy()


def surf_depth_np(allmap, depth_ratio=0.0):
    """twodgs_scene.py:92-111 on numpy arrays: [1,H,W]."""
    alpha = allmap[1:2]
    with np.errstate(divide="ignore", invalid="ignore"):
        exp = np.nan_to_num(allmap[0:1] / alpha, nan=0.0, posinf=0.0, neginf=0.0)
    med = np.nan_to_num(allmap[5:6], nan=0.0, posinf=0.0, neginf=0.0)
    return (exp * (1 - depth_ratio) + depth_ratio * med).astype(np.float32)


def surf_depth_torch(allmap, depth_ratio=0.0):
    import torch
    exp = torch.nan_to_num(allmap[0:1] / allmap[1:2], 0, 0, 0)
    med = torch.nan_to_num(allmap[5:6], 0, 0, 0)
    return exp * (1 - depth_ratio) + depth_ratio * med


def compare_units(got, ref, tsdf_tol=1e-4, color_tol=1.0, bad_frac=1e-3, unit_slack=0.005):
    """got / ref: (coords [n,3], tsdf [n,16,16,16], weight, color [...,3]).  Same units opened (up to `unit_slack` of them at the truncation band's
    edge), and on the common units: weights exact, tsdf within tsdf_tol, colours within color_tol (0..255 scale) on all but `bad_frac` of the voxels."""
    co, t, w, c = got
    rco, rt, rw, rc = ref
    a = {tuple(k): i for i, k in enumerate(np.asarray(co).tolist())}
    b = {tuple(k): i for i, k in enumerate(np.asarray(rco).tolist())}
    common = sorted(set(a) & set(b))
    sym = len(set(a) ^ set(b))
    assert sym <= max(2, unit_slack * len(b)), (len(a), len(b), sym)
    nbad = ntot = 0
    for k in common:
        i, j = a[k], b[k]
        same = w[i] == rw[j]
        nbad += int((~same).sum()); ntot += w[i].size
        nbad += int((np.abs(t[i][same] - rt[j][same]) > tsdf_tol).sum())
        nbad += int((np.abs(c[i][same] - rc[j][same]).max(-1) > color_tol).sum())
    assert nbad <= bad_frac * ntot, (nbad, ntot)
    return dict(units=len(a), units_ref=len(b), units_sym_diff=sym, voxels=ntot, voxels_off=nbad)
