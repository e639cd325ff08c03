"""Fused image-side loss kernels that run right behind the rasterizer."""
import torch

from . import lib, check, ptr, stream_ptr, dev_f32


class _L1PlusLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, gt, aux, waux, root=False):
        c = dev_f32(color, "color", allow_empty=False)
        g = dev_f32(gt, "gt", allow_empty=False)
        a = dev_f32(aux, "aux") if aux is not None else None
        w = dev_f32(waux, "waux") if waux is not None else None
        if (a is None) != (w is None) or (a is not None and a.numel() != w.numel()):
            raise RuntimeError("aux and waux must both be given and have the same number of elements")
        dcol = torch.empty_like(c)
        loss = torch.zeros((), dtype=torch.float32, device=c.device)
        check(lib().gsr_loss_l1_linear(c.numel(), ptr(c), ptr(g), ptr(dcol), a.numel() if a is not None else 0, ptr(a), ptr(w),
                                       ptr(loss), stream_ptr(c.device)), "loss_l1_linear")
        ctx.save_for_backward(dcol, w if w is not None else torch.empty(0, device=c.device))
        ctx.has_aux = a is not None
        ctx.root = bool(root)
        ctx.aux_shape = aux.shape if aux is not None else None
        return loss

    @staticmethod
    def backward(ctx, g):
        dcol, w = ctx.saved_tensors
        if ctx.root:       # the caller declared this value the root of the backward pass: its upstream gradient is exactly 1
            return dcol, None, w.reshape(ctx.aux_shape) if ctx.has_aux else None, None, None
        return dcol * g, None, (w.reshape(ctx.aux_shape) * g) if ctx.has_aux else None, None, None


def l1_plus_linear(color, gt, aux=None, waux=None, root=False):
    """mean|color - gt| + sum(aux * waux), forward value and dL/dcolor in ONE streaming HIP kernel.
    Equivalent torch: (color - gt).abs().mean() + (aux * waux).sum().
    root=True: the returned value is (a plain summand of) the scalar `.backward()` is called on, so its upstream gradient is exactly 1 and
    the stored gradients are handed to autograd as they are -- saves two N-sized multiplies by one per iteration.  Do not set it if the
    loss is rescaled afterwards."""
    return _L1PlusLinear.apply(color, gt, aux, waux, root)


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        x = dev_f32(image, "image", allow_empty=False)
        y = dev_f32(gt, "gt", allow_empty=False)
        if x.dim() != 3 or x.shape != y.shape:
            raise RuntimeError("image and gt must both be (C, H, W)")
        Cc, H, W = x.shape
        L = lib()
        out = torch.empty(3, dtype=torch.float32, device=x.device)
        dimg = torch.empty_like(x)
        scratch = torch.empty(L.gsr_loss_l1_ssim_scratch_bytes(Cc, H, W), dtype=torch.uint8, device=x.device)
        check(L.gsr_loss_l1_ssim(Cc, H, W, ptr(x), ptr(y), float(lambda_dssim), ptr(out), ptr(dimg), ptr(scratch), scratch.numel(),
                                 stream_ptr(x.device)), "loss_l1_ssim")
        ctx.save_for_backward(dimg)
        parts = out[:2].detach()
        ctx.mark_non_differentiable(parts)
        return out[2], parts

    @staticmethod
    def backward(ctx, g, _g_parts):
        (dimg,) = ctx.saved_tensors
        return dimg * g, None, None


def l1_ssim(image, gt, lambda_dssim=0.2, return_parts=False):
    """(1 - lambda) * |image - gt|.mean() + lambda * (1 - ssim(image, gt)) -- the sum of the reference's `L1_loss` and `ssim_loss`
    entries (gssr/scene/vanilla_scene.py:63-69) -- with value and dL/dimage from two fused HIP kernels.  `return_parts=True` also returns
    the (non-differentiable) tensor [mean|image-gt|, mean SSIM] for logging."""
    loss, parts = _L1SSIM.apply(image, gt, lambda_dssim)
    return (loss, parts) if return_parts else loss


def camera_ray_matrices(world_view_transform, full_proj_transform, W, H):
    """-> (ray_mat, normal_rot), 3x3 each, built with the reference's own op sequence (gssr/utils/point_utils.py:9-22,
    gssr/scene/twodgs_scene.py:93): rays_d = [x y 1] @ ray_mat, normal_world = normal_view @ normal_rot.  Per camera, cacheable."""
    wvt, fpt = world_view_transform, full_proj_transform
    c2w = (wvt.T).inverse()
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W) / 2], [0, H / 2, 0, (H) / 2], [0, 0, 0, 1]], dtype=wvt.dtype, device=wvt.device).T
    intrins = ((c2w.T @ fpt) @ ndc2pix)[:3, :3].T
    return (intrins.inverse().T @ c2w[:3, :3].T).contiguous(), wvt[:3, :3].T.contiguous()


class _SurfelGeo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, ray_mat, normal_rot, depth_ratio, lambda_normal, lambda_dist, want_maps):
        am = dev_f32(allmap, "allmap", allow_empty=False)
        if am.dim() != 3 or am.shape[0] != 11:
            raise RuntimeError("allmap must be (11, H, W)")
        rm = dev_f32(ray_mat, "ray_mat", allow_empty=False); nr = dev_f32(normal_rot, "normal_rot", allow_empty=False)
        _, H, W = am.shape
        L = lib()
        dev = am.device
        out = torch.empty(3, dtype=torch.float32, device=dev)
        dL = torch.empty_like(am)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev) if want_maps else None
        nw = torch.empty(3, H, W, dtype=torch.float32, device=dev) if want_maps else None
        sn = torch.empty(3, H, W, dtype=torch.float32, device=dev) if want_maps else None
        scratch = torch.empty(max(L.gsr_loss_surfel_geo_scratch_bytes(H, W), 8), dtype=torch.uint8, device=dev)
        check(L.gsr_loss_surfel_geo(H, W, ptr(am), ptr(rm), ptr(nr), float(depth_ratio), float(lambda_normal), float(lambda_dist), ptr(out),
                                    ptr(dL), ptr(depth), ptr(nw), ptr(sn), ptr(scratch), scratch.numel(), stream_ptr(dev)), "loss_surfel_geo")
        ctx.save_for_backward(dL)
        parts = out[:2].detach()
        extras = [parts] + ([depth, nw, sn] if want_maps else [])
        ctx.mark_non_differentiable(*extras)
        return (out[2], *extras)

    @staticmethod
    def backward(ctx, g, *_):
        (dL,) = ctx.saved_tensors
        return dL * g, None, None, None, None, None, None


def surfel_geo_loss(allmap, ray_mat, normal_rot, depth_ratio=0.0, lambda_normal=0.05, lambda_dist=0.0, return_maps=False):
    """lambda_normal * normal_loss + lambda_dist * dist_loss of TwoDGSScene.get_loss_dict (gssr/scene/twodgs_scene.py:25-35) computed
    straight from the rasterizer's allmap, fused with the render() post-processing (:88-115) and depth_to_normal.
    -> loss, parts=[mean normal error, mean distortion]  (+ depth (1,H,W), normal (3,H,W), surf_normal (3,H,W) when return_maps)."""
    return _SurfelGeo.apply(allmap, ray_mat, normal_rot, depth_ratio, lambda_normal, lambda_dist, bool(return_maps))


class _PlaneGeo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plane_depth, out_all_map, weight, ray_mat, lambda_normal, want_map):
        d = dev_f32(plane_depth, "plane_depth", allow_empty=False)
        am = dev_f32(out_all_map, "out_all_map", allow_empty=False)
        if am.dim() != 3 or am.shape[0] != 5 or d.numel() != am.shape[1] * am.shape[2]:
            raise RuntimeError("out_all_map must be (5, H, W) and plane_depth (1, H, W) / (H, W)")
        w = dev_f32(weight, "weight") if weight is not None else None
        rm = dev_f32(ray_mat, "ray_mat", allow_empty=False)
        _, H, W = am.shape
        L = lib()
        dev = am.device
        out = torch.empty(3, dtype=torch.float32, device=dev)
        dD = torch.empty_like(d)
        dA = torch.zeros_like(am)                       # channels 3 (alpha, detached) and 4 (distance) get no gradient here
        dn = torch.empty(3, H, W, dtype=torch.float32, device=dev) if want_map else None
        scratch = torch.empty(max(L.gsr_loss_surfel_geo_scratch_bytes(H, W), 8), dtype=torch.uint8, device=dev)
        alpha = am[3]
        check(L.gsr_loss_plane_geo(H, W, ptr(d), ptr(alpha), ptr(am), ptr(w), ptr(rm), float(lambda_normal), ptr(out), ptr(dD), ptr(dA), ptr(dn),
                                   ptr(scratch), scratch.numel(), stream_ptr(dev)), "loss_plane_geo")
        ctx.save_for_backward(dD, dA)
        extras = [out[:1].detach()] + ([dn] if want_map else [])
        ctx.mark_non_differentiable(*extras)
        return (out[2], *extras)

    @staticmethod
    def backward(ctx, g, *_):
        dD, dA = ctx.saved_tensors
        return dD * g, dA * g, None, None, None, None


def plane_geo_loss(plane_depth, out_all_map, ray_mat, weight=None, lambda_normal=0.015, return_map=False):
    """PGSR single-view normal loss (gssr/scene/pgsr_scene.py:105-112): lambda * mean(weight * |depth_normal - rendered_normal|.sum(0)) with
    depth_normal = normal_from_depth_image(plane_depth) * alpha.detach(), straight from the rasterizer outputs.  ray_mat = inverse(K^T) of
    `get_calib_matrix_nerf`; weight = the detached image-gradient weight map (per camera, cacheable) or None.
    -> loss, [mean weighted L1]  (+ depth_normal (3,H,W) when return_map)."""
    return _PlaneGeo.apply(plane_depth, out_all_map, weight, ray_mat, lambda_normal, bool(return_map))
