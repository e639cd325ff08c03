"""Synthetic scenes for bench.py, the tools and the parity tests.  numpy only; deterministic per seed.

Camera conventions restate gssr/cameras/__init__.py:85-88 and gssr/utils/graphics_utils.py:38-71 (row-vector
convention: p_view = [p 1] @ world_view_transform); they are pinned against the reference's own helpers by
tests/golden/camera_*.npz (see tests/golden/make_golden.py).
"""
import math
import numpy as np


def projection_matrix(znear, zfar, fovX, fovY):
    """gssr/utils/graphics_utils.py:51-71 (getProjectionMatrix), float32 like torch.zeros(4,4)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world2view(R, t):
    """gssr/utils/graphics_utils.py:38-49 (getWorld2View2 with translate=0, scale=1)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def make_camera(W, H, fx, fy, yaw_deg=0.0, t=(0.0, 0.0, 0.0), znear=0.01, zfar=100.0):
    fovx = 2 * math.atan(W / (2 * fx))
    fovy = 2 * math.atan(H / (2 * fy))
    a = math.radians(yaw_deg)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float64)
    wvt = world2view(R, np.asarray(t, np.float64)).T.copy()                    # cameras/__init__.py:85
    proj = projection_matrix(znear, zfar, fovx, fovy).T.copy()                  # :86
    full = (wvt.astype(np.float32) @ proj.astype(np.float32)).astype(np.float32)  # :87
    center = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)    # :88
    return dict(W=W, H=H, tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                viewmatrix=wvt.astype(np.float32), projmatrix=full, campos=center, fx=fx, fy=fy)


def _quat_to_rot(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    return R


def make_scene(variant, P, W, H, fx=None, fy=None, seed=0, color_mode="precomp", sh_degree=3, pose=0,
               sigma_px=4.0, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, normalize_quat=True, sh_M=16):
    """variant in {'ewa','surfel','plane'}.  Returns the kwargs dict gsrast.runner (and the tests' CPU checker) consume.

    Distribution follows SURVEY.md §8d: z~U[1,20], x,y inside 1.1x the frustum, pixel sigma ~ LogNormal(ln sigma_px, 0.6),
    per-axis factor U[0.3,1] (third axis x0.1 for 'plane', two axes for 'surfel'), opacity sigmoid(N(0,1.5)).
    """
    rng = np.random.default_rng(seed)
    if fx is None:
        fx = W * (1600.0 / 1920.0)
    if fy is None:
        fy = fx
    cam = make_camera(W, H, fx, fy, yaw_deg=(20.0 if pose else 0.0), t=((0.5, 0.2, 0.0) if pose else (0, 0, 0)))
    z = rng.uniform(1.0, 20.0, P)
    x = z * cam["tanfovx"] * rng.uniform(-1.1, 1.1, P)
    y = z * cam["tanfovy"] * rng.uniform(-1.1, 1.1, P)
    pts_cam = np.stack([x, y, z], -1)
    # camera -> world with the row-vector convention: p_view = p_world @ V[:3,:3] + V[3,:3]
    V = cam["viewmatrix"].astype(np.float64)
    means3D = (pts_cam - V[3, :3]) @ np.linalg.inv(V[:3, :3])
    spx = np.exp(rng.normal(math.log(sigma_px), 0.6, P))
    s = spx * z / fx
    nax = 2 if variant == "surfel" else 3
    scales = s[:, None] * rng.uniform(0.3, 1.0, (P, nax))
    if variant == "plane":
        scales[:, 2] *= 0.1
    q = rng.normal(0, 1, (P, 4))
    if normalize_quat:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0, 1.5, P)))
    sc = dict(cam)
    sc.update(variant=variant, means3D=means3D.astype(np.float32), scales=scales.astype(np.float32),
              rotations=q.astype(np.float32), opacities=opac.astype(np.float32)[:, None],
              bg=np.asarray(bg, np.float32), scale_modifier=scale_modifier, sh_degree=0, render_geo=True)
    if color_mode == "sh":
        M = int(sh_M)                # 16 = what every reference model allocates ((max_sh_degree + 1)^2, base_gaussian.py); 1 / 4 / 9 = models
        assert (sh_degree + 1) ** 2 <= M     # built with max_sh_degree 0 / 1 / 2 (the kernels take M = shs.size(1), rasterize_points.cu:66-70)
        shs = rng.normal(0, 0.1, (P, M, 3))
        shs[:, 0, :] = rng.uniform(-1, 1, (P, 3)) / 0.28209479177387814
        sc["shs"] = shs.astype(np.float32)
        sc["sh_degree"] = sh_degree
    else:
        sc["colors_precomp"] = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    if variant == "plane":
        sc["all_map"] = plane_all_map(sc)
    return sc


def plane_all_map(sc):
    """Restates gssr/scene/pgsr_scene.py:245-302 (smallest-axis normal flipped towards the camera, plane distance)."""
    q = sc["rotations"].astype(np.float64)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)      # pytorch3d quaternion_to_matrix normalises via 2/|q|^2
    R = _quat_to_rot(q)
    idx = np.argmin(sc["scales"], axis=1)
    normal = R[np.arange(R.shape[0]), :, idx]
    to_cam = sc["campos"].astype(np.float64)[None] - sc["means3D"].astype(np.float64)
    neg = (normal * to_cam).sum(-1) < 0
    normal[neg] = -normal[neg]
    V = sc["viewmatrix"].astype(np.float64)
    local_n = normal @ V[:3, :3]
    pts_cam = sc["means3D"].astype(np.float64) @ V[:3, :3] + V[3, :3]
    dist = np.abs((local_n * pts_cam).sum(-1))
    am = np.zeros((R.shape[0], 5), np.float32)
    am[:, :3] = local_n
    am[:, 3] = 1.0
    am[:, 4] = dist
    return am


def random_out_grads(variant, W, H, seed=0, scale=None):
    """Upstream gradients dL/d(outputs) ~ N(0,1)/N."""
    rng = np.random.default_rng(1000 + seed)
    n = W * H
    s = (1.0 / n) if scale is None else scale
    g = dict(dL_dcolor=(rng.normal(0, 1, (3, H, W)) * s).astype(np.float32))
    if variant == "surfel":
        o = (rng.normal(0, 1, (11, H, W)) * s).astype(np.float32)
        o[7] = 0.0     # median idx is an integer output, no gradient
        g["dL_dothers"] = o
    if variant == "plane":
        g["dL_dout_all_map"] = (rng.normal(0, 1, (5, H, W)) * s).astype(np.float32)
        g["dL_dplane_depth"] = (rng.normal(0, 1, (1, H, W)) * s).astype(np.float32)
    return g


def concentrate(sc, frac, scale):
    """Pull the first `frac` of the gaussians towards the optical axis (camera-space x, y scaled by `scale`): object-centric density, a few hundred
    tiles with lists several thousand entries long.  For the long-list paths of the per-tile depth sort and of the blend backward;
    not the BASELINE workload."""
    V = sc["viewmatrix"].astype(np.float64)
    n = int(frac * sc["means3D"].shape[0])
    pc = sc["means3D"][:n].astype(np.float64) @ V[:3, :3] + V[3, :3]
    pc[:, :2] *= scale
    sc["means3D"][:n] = ((pc - V[3, :3]) @ np.linalg.inv(V[:3, :3])).astype(np.float32)
    if sc.get("all_map") is not None:
        sc["all_map"] = plane_all_map(sc)
    return sc
