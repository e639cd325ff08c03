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


# ---- PGSR: normal_from_depth_image chain (gssr/utils/graphics_utils.py:80-146) and the single-view normal loss (pgsr_scene.py:105-112,320)
def ndc_2_cam(ndc_xyz, intrinsic, W, H):
    inv_scale = torch.tensor([[W - 1, H - 1]], device=ndc_xyz.device, dtype=ndc_xyz.dtype)
    cam_z = ndc_xyz[..., 2:3]
    cam_xy = ndc_xyz[..., :2] * inv_scale * cam_z
    cam_xyz = torch.cat([cam_xy, cam_z], dim=-1)
    return cam_xyz @ torch.inverse(intrinsic[0, ...].t())


def depth2point_cam(sampled_depth, ref_intrinsic):
    B, N, C, H, W = sampled_depth.shape
    valid_z = sampled_depth
    valid_x = torch.arange(W, dtype=sampled_depth.dtype, device=sampled_depth.device) / (W - 1)
    valid_y = torch.arange(H, dtype=sampled_depth.dtype, device=sampled_depth.device) / (H - 1)
    valid_y, valid_x = torch.meshgrid(valid_y, valid_x, indexing="ij")
    valid_x = valid_x[None, None, None, ...].expand(B, N, C, -1, -1)
    valid_y = valid_y[None, None, None, ...].expand(B, N, C, -1, -1)
    ndc_xyz = torch.stack([valid_x, valid_y, valid_z], dim=-1).view(B, N, C, H, W, 3)
    return ndc_xyz, ndc_2_cam(ndc_xyz, ref_intrinsic, W, H)


def depth_pcd2normal(xyz):
    hd, wd, _ = xyz.shape
    bottom_point = xyz[..., 2:hd, 1:wd - 1, :]
    top_point = xyz[..., 0:hd - 2, 1:wd - 1, :]
    right_point = xyz[..., 1:hd - 1, 2:wd, :]
    left_point = xyz[..., 1:hd - 1, 0:wd - 2, :]
    left_to_right = right_point - left_point
    bottom_to_top = top_point - bottom_point
    xyz_normal = torch.cross(left_to_right, bottom_to_top, dim=-1)
    xyz_normal = torch.nn.functional.normalize(xyz_normal, p=2, dim=-1)
    return torch.nn.functional.pad(xyz_normal.permute(2, 0, 1), (1, 1, 1, 1), mode='constant').permute(1, 2, 0)


def normal_from_depth_image(depth, intrinsic_matrix):
    _, xyz_cam = depth2point_cam(depth[None, None, None, ...], intrinsic_matrix[None, ...])
    return depth_pcd2normal(xyz_cam.reshape(*depth.shape, 3))


def plane_geo_loss(plane_depth, out_all_map, K, weight, lambda_normal):
    """plane_depth (H,W); out_all_map (5,H,W); K (3,3).  -> loss, mean weighted L1, depth_normal (3,H,W)"""
    rendered_normal, rendered_alpha = out_all_map[0:3], out_all_map[3:4]
    depth_normal = normal_from_depth_image(plane_depth, K).permute(2, 0, 1) * rendered_alpha.detach()
    w = torch.ones_like(plane_depth) if weight is None else weight
    m = (w * ((depth_normal - rendered_normal).abs().sum(0))).mean()
    return lambda_normal * m, m, depth_normal
