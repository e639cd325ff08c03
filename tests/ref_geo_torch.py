"""Autograd reference for the geometric regularisers, written from the formulas (TEST INFRASTRUCTURE ONLY).

What is evaluated (the definitions live in the reference at gssr/scene/twodgs_scene.py:25-35,88-115, gssr/utils/point_utils.py:9-37,
gssr/utils/graphics_utils.py:80-146, gssr/scene/pgsr_scene.py:105-112,320; their outputs are pinned as data in
tests/golden/ref_loss_surfel_geo_r{0,1}.npz and ref_loss_plane_geo.npz):

  2DGS   surface depth  D = (1-r) * E/alpha + r * Dmed          (E expected depth, nan -> 0)
         world point    X(x, y) = o + D(x, y) * ([x y 1] M)       (M = K^-1 then camera-to-world rotation, o = camera centre)
         depth normal   n = normalize((X[y+1] - X[y-1]) x (X[x+1] - X[x-1])) on interior pixels, 0 on the border, times alpha (detached)
         loss           lambda_n * mean(1 - <N_world, n>) + lambda_d * mean(distortion)
  PGSR   camera point   Xc(x, y) = z * K^-1 [x y 1]
         depth normal   n = normalize((Xc[x+1] - Xc[x-1]) x (Xc[y-1] - Xc[y+1])), 0 on the border, times alpha (detached)
         loss           lambda_n * mean(w * |n - N|_1)

Everything is plain broadcasting over an (H, W) pixel lattice in the dtype / on the device of the inputs, so float64 autograd of these
functions is the truth the C oracle's analytic gradients are held to (tests/test_loss_cpu.py)."""
import torch
import torch.nn.functional as F


def _lattice(H, W, like):
    """Homogeneous pixel coordinates [x, y, 1] as an (H, W, 3) tensor."""
    ys = torch.arange(H, dtype=like.dtype, device=like.device).view(H, 1).expand(H, W)
    xs = torch.arange(W, dtype=like.dtype, device=like.device).view(1, W).expand(H, W)
    return torch.stack((xs, ys, torch.ones_like(xs)), dim=2)


def _central_normals(X, flip):
    """X: (H, W, 3) point map.  Unit normal from central differences on interior pixels, zero on the one-pixel border.
    flip=False: (down - up) x (right - left)  [2DGS orientation];  flip=True: (right - left) x (up - down)  [PGSR orientation]."""
    H, W, _ = X.shape
    out = X * 0                                   # stays in the graph: an image without interior pixels still has a (zero) gradient
    if H < 3 or W < 3:
        return out
    along_y = X[2:, 1:-1] - X[:-2, 1:-1]          # down minus up
    along_x = X[1:-1, 2:] - X[1:-1, :-2]          # right minus left
    n = torch.linalg.cross(along_x, -along_y, dim=2) if flip else torch.linalg.cross(along_y, along_x, dim=2)
    out[1:-1, 1:-1] = F.normalize(n, dim=2)
    return out


def ray_matrices(world_view_transform, full_proj_transform, W, H):
    """-> (M, Rn): world ray of pixel (x, y) = [x y 1] @ M;  N_world = N_view @ Rn.  Matrices are in the row-vector convention GS-SR stores
    (world_view_transform = [R|t]^T, full_proj_transform = world_view_transform @ projection)."""
    V = world_view_transform
    proj = torch.linalg.solve(V, full_proj_transform)               # V^-1 (V P) = P: the projection alone
    fx, fy = proj[0, 0] * (W / 2), proj[1, 1] * (H / 2)
    cx, cy = (proj[2, 0] + 1) * (W / 2), (proj[2, 1] + 1) * (H / 2)            # proj[2, 3] = +1: the camera looks down +z
    one, zero = torch.ones_like(fx), torch.zeros_like(fx)
    Kinv_T = torch.stack((torch.stack((1 / fx, zero, zero)), torch.stack((zero, 1 / fy, zero)), torch.stack((-cx / fx, -cy / fy, one))))
    cam_to_world = torch.linalg.inv(V)                               # row-vector convention: x_world = x_cam @ cam_to_world
    return Kinv_T @ cam_to_world[:3, :3], V[:3, :3].T


def geo_loss(allmap, wvt, fpt, depth_ratio, lambda_normal, lambda_dist):
    """allmap (11, H, W) as the surfel rasterizer returns it -> (loss, normal term, distortion term, maps)."""
    _, H, W = allmap.shape
    M, Rn = ray_matrices(wvt, fpt, W, H)
    centre = torch.linalg.inv(wvt)[3, :3]
    alpha = allmap[1]
    expected = torch.nan_to_num(allmap[0] / alpha, nan=0.0, posinf=0.0, neginf=0.0)
    median = torch.nan_to_num(allmap[5], nan=0.0, posinf=0.0, neginf=0.0)
    D = (1 - depth_ratio) * expected + depth_ratio * median
    X = centre + D.unsqueeze(2) * (_lattice(H, W, allmap) @ M)
    n_depth = _central_normals(X, flip=False) * alpha.detach().unsqueeze(2)                  # (H, W, 3)
    n_world = torch.einsum("chw,cd->hwd", allmap[2:5], Rn)
    err = (1 - (n_world * n_depth).sum(dim=2)).mean()
    dist = allmap[6].mean()
    maps = {"rend_alpha": alpha.unsqueeze(0), "rend_dist": allmap[6:7], "surf_normal": n_depth.permute(2, 0, 1), "depth": D.unsqueeze(0),
            "normal": n_world.permute(2, 0, 1)}
    return lambda_normal * err + lambda_dist * dist, err, dist, maps


def plane_geo_loss(plane_depth, out_all_map, K, weight, lambda_normal):
    """plane_depth (H, W); out_all_map (5, H, W): rendered normal 0:3, alpha 3; K (3, 3) -> (loss, mean weighted L1, depth normal (3, H, W))."""
    H, W = plane_depth.shape
    Xc = plane_depth.unsqueeze(2) * (_lattice(H, W, plane_depth) @ torch.linalg.inv(K).T)
    n = _central_normals(Xc, flip=True).permute(2, 0, 1) * out_all_map[3:4].detach()
    l1 = (n - out_all_map[0:3]).abs().sum(dim=0)
    m = (l1 if weight is None else weight * l1).mean()
    return lambda_normal * m, m, n
