"""Transcription of the 2DGS geometric post-processing and regularisers: TwoDGSScene.render (gssr/scene/twodgs_scene.py:88-115),
depths_to_points / depth_to_normal (gssr/utils/point_utils.py:9-37), get_loss_dict (twodgs_scene.py:25-35).  dtype/device selectable;
the hard-coded .cuda() calls of the reference are replaced by the input's device.  TEST INFRASTRUCTURE ONLY."""
import torch


def ray_matrices(world_view_transform, full_proj_transform, W, H):
    """-> (ray_mat 3x3: rays_d = [x y 1] @ ray_mat,  normal_rot 3x3: n_world = n_view @ normal_rot) exactly as the reference forms them."""
    wvt, fpt = world_view_transform, full_proj_transform
    c2w = (wvt.T).inverse()
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W) / 2], [0, H / 2, 0, (H) / 2], [0, 0, 0, 1]], dtype=wvt.dtype, device=wvt.device).T
    projection_matrix = c2w.T @ fpt
    intrins = (projection_matrix @ ndc2pix)[:3, :3].T
    return intrins.inverse().T @ c2w[:3, :3].T, wvt[:3, :3].T


def depths_to_points(wvt, fpt, W, H, depthmap):
    c2w = (wvt.T).inverse()
    ray_mat, _ = ray_matrices(wvt, fpt, W, H)
    grid_x, grid_y = torch.meshgrid(torch.arange(W, device=wvt.device).to(wvt.dtype), torch.arange(H, device=wvt.device).to(wvt.dtype), indexing='xy')
    points = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)], dim=-1).reshape(-1, 3)
    rays_d = points @ ray_mat
    rays_o = c2w[:3, 3]
    return depthmap.reshape(-1, 1) * rays_d + rays_o


def depth_to_normal(wvt, fpt, W, H, depth):
    points = depths_to_points(wvt, fpt, W, H, depth).reshape(*depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = torch.cat([points[2:, 1:-1] - points[:-2, 1:-1]], dim=0)
    dy = torch.cat([points[1:-1, 2:] - points[1:-1, :-2]], dim=1)
    normal_map = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    output[1:-1, 1:-1, :] = normal_map
    return output


def render_post(allmap, wvt, fpt, depth_ratio):
    _, H, W = allmap.shape
    render_alpha = allmap[1:2]
    render_normal = allmap[2:5]
    render_normal = (render_normal.permute(1, 2, 0) @ (wvt[:3, :3].T)).permute(2, 0, 1)
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    render_dist = allmap[6:7]
    surf_depth = render_depth_expected * (1 - depth_ratio) + (depth_ratio) * render_depth_median
    surf_normal = depth_to_normal(wvt, fpt, W, H, surf_depth).permute(2, 0, 1)
    surf_normal = surf_normal * (render_alpha).detach()
    return {"rend_alpha": render_alpha, "rend_dist": render_dist, "surf_normal": surf_normal, "depth": surf_depth, "normal": render_normal}


def geo_loss(allmap, wvt, fpt, depth_ratio, lambda_normal, lambda_dist):
    o = render_post(allmap, wvt, fpt, depth_ratio)
    normal_error = (1 - (o["normal"] * o["surf_normal"]).sum(dim=0))[None]
    return lambda_normal * normal_error.mean() + lambda_dist * o["rend_dist"].mean(), normal_error.mean(), o["rend_dist"].mean(), o
