"""Fused image-side loss kernels that run right behind the rasterizer."""
import threading

import torch

from . import lib, check, ptr, stream_ptr, dev_f32


# Zero-initialised scalars for kernels that accumulate a loss value with atomics: one torch.zeros(1024) serves 1024 calls (a fill kernel per
# call costs ~4 us + a launch gap on a 1 ms iteration).  Every scalar is handed out once; a spent block stays alive through its views.
# The pool is keyed by (device, current stream) -- a block is filled on the stream that allocates it and its scalars are only handed to work on that
# stream, so the fill is ordered before every use -- and guarded by a lock: two Python threads never receive the same word.
_ZEROS = {}
_ZEROS_LOCK = threading.Lock()


def zero_scalar(device):
    if torch.cuda.is_current_stream_capturing():          # inside a graph the fill has to be part of the graph: replays re-zero it
        return torch.zeros((), dtype=torch.float32, device=device)
    device = torch.device(device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    with _ZEROS_LOCK:
        ent = _ZEROS.get(key)
        if ent is None or ent[1] >= ent[0].numel():
            ent = _ZEROS[key] = [torch.zeros(1024, dtype=torch.float32, device=device), 0]
        v = ent[0][ent[1]]
        ent[1] += 1
    return v


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
        loss = zero_scalar(c.device)
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


class _ScalingProd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, lam, cols, count, unit_upstream):
        s = dev_f32(scaling, "scaling", allow_empty=False)
        if s.dim() != 2 or not (1 <= cols <= min(3, s.shape[1])):
            raise RuntimeError("scaling_prod_mean: scaling must be (P, >= cols) with cols in 1..3")
        if count is not None and (count.dtype != torch.int32 or not count.is_cuda):
            raise RuntimeError("scaling_prod_mean: count must be a device int32 tensor")
        grad = torch.empty_like(s)
        loss = zero_scalar(s.device)
        check(lib().gsr_loss_scaling_prod(s.shape[0], int(cols), s.shape[1], ptr(s), ptr(count) if count is not None else None, float(lam), ptr(loss),
                                          ptr(grad), stream_ptr(s.device)), "loss_scaling_prod")
        ctx.save_for_backward(grad)
        ctx.unit = bool(unit_upstream)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad if ctx.unit else grad * g), None, None, None, None


def scaling_prod_mean(scaling, lambda_scaling=1.0, cols=None, count=None, unit_upstream=False):
    """`lambda_scaling * scaling[:, :cols].prod(dim=1).mean()` (the scaling_loss of the scaffold / octree scenes, scaffold_scene.py:184 and its
    2DGS / PGSR variants) -- value and gradient in one HIP kernel instead of torch's slice / prod / mean chain and its backward (7-9 launches; prod's
    backward also synchronises the host when a scale is exactly 0).  cols: leading columns in the product (default: all; 2DGS uses 2 of the decode's 3).
    count: device int32 [1] divisor instead of the number of rows (static-shape iterations).  unit_upstream=True: the value is a plain summand of the
    scalar .backward() is called on (upstream gradient exactly 1), the stored gradient is handed to autograd as it is."""
    return _ScalingProd.apply(scaling, float(lambda_scaling), int(cols if cols is not None else scaling.shape[1]), count, unit_upstream)


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim, unit_upstream=False):
        ctx.unit = bool(unit_upstream)
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
        ctx.set_materialize_grads(False)
        return out[2], parts

    @staticmethod
    def backward(ctx, g, _g_parts):
        (dimg,) = ctx.saved_tensors
        return (None if g is None else (dimg if ctx.unit else dimg * g)), None, None, None


def l1_ssim(image, gt, lambda_dssim=0.2, return_parts=False, unit_upstream=False):
    """(1 - lambda) * |image - gt|.mean() + lambda * (1 - ssim(image, gt)) -- the sum of the reference's `L1_loss` and `ssim_loss`
    entries (gssr/scene/vanilla_scene.py:63-69) -- with value and dL/dimage from two fused HIP kernels.  `return_parts=True` also returns
    the (non-differentiable) tensor [mean|image-gt|, mean SSIM] for logging.
    `unit_upstream=True` (here and in the geometric losses below) is the caller's promise that this value enters the total loss with weight
    exactly 1 -- as every entry of GS-SR's `sum(loss_dict.values())` does -- so that backward hands out the gradient computed in forward as
    it is instead of multiplying a full-size map by the upstream scalar 1."""
    loss, parts = _L1SSIM.apply(image, gt, lambda_dssim, bool(unit_upstream))
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
    def forward(ctx, allmap, ray_mat, normal_rot, depth_ratio, lambda_normal, lambda_dist, want_maps, unit_upstream=False):
        ctx.unit = bool(unit_upstream)
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
        ctx.set_materialize_grads(False)     # the extra outputs are plain values: no zero-filled (3,H,W) gradients for them on every backward
        return (out[2], *extras)

    @staticmethod
    def backward(ctx, g, *_):
        (dL,) = ctx.saved_tensors
        return (None if g is None else (dL if ctx.unit else dL * g)), None, None, None, None, None, None, None


def surfel_geo_loss(allmap, ray_mat, normal_rot, depth_ratio=0.0, lambda_normal=0.05, lambda_dist=0.0, return_maps=False, unit_upstream=False):
    """lambda_normal * normal_loss + lambda_dist * dist_loss of TwoDGSScene.get_loss_dict (gssr/scene/twodgs_scene.py:25-35) computed
    straight from the rasterizer's allmap, fused with the render() post-processing (:88-115) and depth_to_normal.
    -> loss, parts=[mean normal error, mean distortion]  (+ depth (1,H,W), normal (3,H,W), surf_normal (3,H,W) when return_maps)."""
    return _SurfelGeo.apply(allmap, ray_mat, normal_rot, depth_ratio, lambda_normal, lambda_dist, bool(return_maps), bool(unit_upstream))


def _plane_geo_fwd(plane_depth, out_all_map, weight, ray_mat, lambda_normal, want_map):
    """gsr_loss_plane_geo: -> (out[3] = {mean weighted L1, -, lambda * mean}, dD, dA (5,H,W; channels 3, 4 zero), depth_normal map or None)."""
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
    dA = torch.empty_like(am)                       # the kernel writes every pixel of the three normal channels;
    dA[3:].zero_()                                  # channels 3 (alpha, detached) and 4 (distance) get no gradient here
    dn = torch.empty(3, H, W, dtype=torch.float32, device=dev) if want_map else None
    scratch = torch.empty(max(L.gsr_loss_surfel_geo_scratch_bytes(H, W), 8), dtype=torch.uint8, device=dev)
    alpha = am[3]
    check(L.gsr_loss_plane_geo(H, W, ptr(d), ptr(alpha), ptr(am), ptr(w), ptr(rm), float(lambda_normal), ptr(out), ptr(dD), ptr(dA), ptr(dn),
                               ptr(scratch), scratch.numel(), stream_ptr(dev)), "loss_plane_geo")
    return out, dD, dA, dn


class _PlaneGeo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plane_depth, out_all_map, weight, ray_mat, lambda_normal, want_map, unit_upstream=False):
        ctx.unit = bool(unit_upstream)
        out, dD, dA, dn = _plane_geo_fwd(plane_depth, out_all_map, weight, ray_mat, lambda_normal, want_map)
        ctx.save_for_backward(dD, dA)
        extras = [out[:1].detach()] + ([dn] if want_map else [])
        ctx.mark_non_differentiable(*extras)
        ctx.set_materialize_grads(False)
        return (out[2], *extras)

    @staticmethod
    def backward(ctx, g, *_):
        dD, dA = ctx.saved_tensors
        if g is None:
            return None, None, None, None, None, None, None
        if ctx.unit:
            return dD, dA, None, None, None, None, None
        return dD * g, dA * g, None, None, None, None, None


def plane_geo_loss(plane_depth, out_all_map, ray_mat, weight=None, lambda_normal=0.015, return_map=False, unit_upstream=False):
    """PGSR single-view normal loss (gssr/scene/pgsr_scene.py:105-112): lambda * mean(weight * |depth_normal - rendered_normal|.sum(0)) with
    depth_normal = normal_from_depth_image(plane_depth) * alpha.detach(), straight from the rasterizer outputs.  ray_mat = inverse(K^T) of
    `get_calib_matrix_nerf`; weight = the detached image-gradient weight map (per camera, cacheable) or None.
    -> loss, [mean weighted L1]  (+ depth_normal (3,H,W) when return_map)."""
    return _PlaneGeo.apply(plane_depth, out_all_map, weight, ray_mat, lambda_normal, bool(return_map), bool(unit_upstream))


def multiview_cfg(view_cam, near_cam, W, H, near_size=None, gray_size=None, patch_size=3, pixel_noise_threshold=1.0):
    """gsr_mv_cfg from two reference Cameras (attributes R, T, Fx, Fy, Cx, Cy, ncc_scale: gssr/cameras/__init__.py:36-88).
    The relative pose is composed on the host in float64 (X_near = X_view A + b), once per camera pair."""
    import numpy as np
    from . import MvCfg
    Rv, Tv, Rn, Tn = (np.asarray(a, dtype=np.float64) for a in (view_cam.R, view_cam.T, near_cam.R, near_cam.T))
    A = Rv.T @ Rn
    b = Tn - Tv @ A
    Ai = Rn.T @ Rv
    bi = Tv - Tn @ Ai
    Wn, Hn = near_size if near_size is not None else (int(near_cam.image_width), int(near_cam.image_height))
    Wg, Hg = gray_size if gray_size is not None else (W, H)
    c = MvCfg(int(W), int(H), int(Wn), int(Hn), int(Wg), int(Hg), float(view_cam.Fx), float(view_cam.Fy), float(view_cam.Cx), float(view_cam.Cy),
              float(near_cam.Fx), float(near_cam.Fy), float(near_cam.Cx), float(near_cam.Cy))
    c.v2n[:] = [float(v) for v in np.concatenate([A.reshape(-1), b])]
    c.n2v[:] = [float(v) for v in np.concatenate([Ai.reshape(-1), bi])]
    c.ncc_scale = float(getattr(view_cam, "ncc_scale", 1.0))
    c.noise_th = float(pixel_noise_threshold)
    c.patch = int(patch_size)
    return c


_sample_calls = 0


def sample_valid_pixels(d_mask, num_sample, generator=None, seed=None):
    """Uniform sample WITHOUT replacement of at most `num_sample` pixels of d_mask (pgsr_scene.py:147-151 draws it with np.random.choice on
    the host), without leaving the device and without a sort (gsr_sample_mask: hashed keys + two-level histogram select + scan compaction).
    -> int32 [min(num_sample, H*W)] pixel indices, ascending, -1 in unused slots (all valid pixels when there are fewer than num_sample).
    `seed` (int) fixes the draw; otherwise it is taken from `generator` (CPU generators are advanced; for device generators initial_seed() is
    mixed with a call counter so that no device->host read is needed) or from torch's default CPU generator."""
    global _sample_calls
    m = d_mask.reshape(-1)
    n = m.numel()
    if n <= num_sample:
        ar = torch.arange(n, dtype=torch.int32, device=m.device)
        return torch.where(m.bool(), ar, torch.full_like(ar, -1))
    if not m.is_cuda:
        raise RuntimeError("sample_valid_pixels: d_mask must be a HIP-device tensor (no CPU path)")
    if seed is None:
        if generator is not None and generator.device.type != "cpu":
            _sample_calls += 1
            seed = (int(generator.initial_seed()) * 0x9E3779B97F4A7C15 + _sample_calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        else:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())          # CPU tensor: no device sync
    m8 = m.contiguous() if m.dtype == torch.uint8 else m.to(torch.uint8).contiguous()
    L = lib()
    out = torch.empty(num_sample, dtype=torch.int32, device=m.device)
    scratch = torch.empty(L.gsr_sample_mask_scratch_bytes(n), dtype=torch.uint8, device=m.device)
    check(L.gsr_sample_mask(n, ptr(m8), int(num_sample), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(out), ptr(scratch), scratch.numel(), stream_ptr(m.device)),
          "sample_mask")
    return out


class _PlaneMultiview(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plane_depth, near_plane_depth, normal, distance, gray, near_gray, cfg, lambda_geo, lambda_ncc, num_sample, indices, generator,
                all_map=None, single_view=None):
        import ctypes as C
        d = dev_f32(plane_depth, "plane_depth", allow_empty=False)
        nd = dev_f32(near_plane_depth, "near_plane_depth", allow_empty=False)
        am = None
        if all_map is not None:                  # normal = channels 0-2, distance = channel 4 of the rasterizer's out_all_map: ONE gradient tensor back
            am = dev_f32(all_map, "out_all_map", allow_empty=False)
            if am.dim() != 3 or am.shape[0] != 5:
                raise RuntimeError("plane_multiview_loss: out_all_map must be (5, H, W)")
            normal, distance = am[0:3], am[4:5]
        nm = dev_f32(normal, "rendered_normal", allow_empty=False)
        ds = dev_f32(distance, "rendered_distance", allow_empty=False)
        g = dev_f32(gray, "gray", allow_empty=False)
        ng = dev_f32(near_gray, "near_gray", allow_empty=False)
        W, H = cfg.W, cfg.H
        if d.numel() != W * H or nd.numel() != cfg.Wn * cfg.Hn or nm.numel() != 3 * W * H or ds.numel() != W * H:
            raise RuntimeError("plane_multiview_loss: map sizes do not match the configuration")
        if g.numel() != cfg.Wg * cfg.Hg or ng.numel() != cfg.Wg * cfg.Hg:
            raise RuntimeError("plane_multiview_loss: both gray images must be (1, Hg, Wg)")
        L = lib()
        dev = d.device
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        noise, weight, gD, gN = f(H * W), f(H * W), f(H * W), f(cfg.Hn * cfg.Wn)
        dmask = torch.empty(H * W, dtype=torch.uint8, device=dev)
        stats = f(6)
        nmax = min(int(num_sample), H * W) if indices is None else int(indices.numel())
        scratch = torch.empty(max(L.gsr_loss_plane_mv_scratch_bytes(W, H, nmax), 8), dtype=torch.uint8, device=dev)
        check(L.gsr_loss_plane_mv_geo(C.byref(cfg), ptr(d), ptr(nd), ptr(noise), ptr(dmask), ptr(weight), ptr(stats), ptr(gD), ptr(gN), ptr(scratch),
                                      scratch.numel(), stream_ptr(dev)), "loss_plane_mv_geo")
        if indices is None:
            indices = sample_valid_pixels(dmask, num_sample, generator)
        idx = indices.to(device=dev, dtype=torch.int32).contiguous()
        if am is not None:
            gAM = f(5, H, W); gAM[3].zero_()
            gNm, gDs = gAM[0:3], gAM[4]
        else:
            gAM = f(4, H, W)                     # normal (3) and distance (1) gradient maps in one buffer: one scaling launch in backward
            gNm, gDs = gAM[0:3], gAM[3]
        ncc = f(max(idx.numel(), 1))
        cmask = torch.empty(max(idx.numel(), 1), dtype=torch.uint8, device=dev)
        check(L.gsr_loss_plane_mv_ncc(C.byref(cfg), int(idx.numel()), ptr(idx), ptr(weight), ptr(nm), ptr(ds), ptr(g), ptr(ng), ptr(ncc), ptr(cmask),
                                      ptr(stats[3:]), ptr(gNm), ptr(gDs), ptr(scratch), scratch.numel(), stream_ptr(dev)), "loss_plane_mv_ncc")
        # loss values lambda * mean on the device, one launch (the means' 1 / count and the lambdas reach the gradient maps in backward, one launch too)
        vals = f(2)
        check(L.gsr_loss_plane_mv_values(ptr(stats), float(lambda_geo), float(lambda_ncc), ptr(vals), stream_ptr(dev)), "loss_plane_mv_values")
        geo, nccl = vals[0], vals[1]
        # single_view = (ray_mat, weight, lambda_normal): the single-view normal loss of the same render (plane_geo_loss) evaluated by THIS node, so that
        # its gradients to plane_depth / out_all_map leave backward already summed with the multi-view ones (plane_losses below)
        sv_val, svD, svA = None, None, None
        if single_view is not None:
            if am is None:
                raise RuntimeError("plane_losses needs out_all_map")
            out3, svD, svA, _ = _plane_geo_fwd(plane_depth, am, single_view[1], single_view[0], single_view[2], False)
            sv_val = out3[2]
        ctx.has_sv = svD is not None
        ctx.save_for_backward(gD, gN, gAM, stats, *([svD, svA] if svD is not None else []))
        ctx.lams = (float(lambda_geo), float(lambda_ncc))
        ctx.whole = am is not None
        ctx.shapes = (plane_depth.shape, near_plane_depth.shape, normal.shape, distance.shape)
        aux = {"pixel_noise": noise.view(H, W), "d_mask": dmask.view(H, W).bool(), "weights": weight.view(H, W), "indices": idx,
               "ncc": ncc[: idx.numel()], "ncc_mask": cmask[: idx.numel()].bool(), "stats": stats}
        ctx.mark_non_differentiable(*[v for v in aux.values()])
        ctx.set_materialize_grads(False)     # seven auxiliary outputs (H*W masks, sample lists): no zero-filled gradients for them
        if sv_val is not None:
            return (geo, nccl, sv_val, *aux.values())
        return (geo, nccl, *aux.values())

    @staticmethod
    def backward(ctx, g_geo, g_ncc, *rest):
        sv = ctx.saved_tensors
        gD, gN, gAM, stats = sv[:4]
        svD, svA = (sv[4], sv[5]) if ctx.has_sv else (None, None)
        g_sv = rest[0] if ctx.has_sv else None
        s0, s1, s2, s3 = ctx.shapes
        lg, ln = ctx.lams
        L = lib()
        dev = gD.device
        up = lambda g: None if g is None else g.reshape(1).to(torch.float32)      # a view for the float32 scalars autograd sends
        ug, un, us = up(g_geo), up(g_ncc), up(g_sv)   # a loss the caller did not use sends no gradient: its maps are not produced
        add = us is not None
        oD = torch.empty_like(gD) if (ug is not None or add) else None
        oN = torch.empty_like(gN) if ug is not None else None
        oA = torch.empty_like(gAM) if (un is not None or add) else None
        check(L.gsr_loss_plane_mv_scale(gD.numel() if oD is not None else 0, gN.numel() if oN is not None else 0, gAM.numel() if oA is not None else 0,
                                        ptr(gD) if ug is not None else None, ptr(gN), ptr(gAM) if un is not None else None, ptr(stats), lg, ln, ptr(ug), ptr(un),
                                        ptr(svD) if add else None, ptr(svA) if add else None, ptr(us), 1 if add else 0,
                                        ptr(oD), ptr(oN), ptr(oA), stream_ptr(dev)), "loss_plane_mv_scale")
        gd = None if oD is None else oD.view(s0)
        gn = None if oN is None else oN.view(s1)
        if ctx.whole:                              # the whole (5,H,W) gradient of out_all_map
            return (gd, gn, None, None, None, None, None, None, None, None, None, None, oA, None)
        return (gd, gn, None if oA is None else oA[0:3].view(s2), None if oA is None else oA[3].view(s3), None, None, None, None, None, None, None,
                None, None, None)


_MV_AUX = ("pixel_noise", "d_mask", "weights", "indices", "ncc", "ncc_mask", "stats")


def plane_multiview_loss(plane_depth, near_plane_depth, rendered_normal, rendered_distance, gray, near_gray, cfg, lambda_geo=0.03, lambda_ncc=0.15,
                         num_sample=102400, indices=None, generator=None, return_aux=False, out_all_map=None):
    """PGSR multi-view losses (gssr/scene/pgsr_scene.py:113-204): -> (geo_loss, ncc_loss) [, aux dict].

    plane_depth (1,H,W) / rendered_normal (3,H,W) / rendered_distance (1,H,W): this view's render; near_plane_depth: `nearest_render_pkg
    ['plane_depth']` (it receives gradient, as in the reference); gray / near_gray: `viewpoint_cam.gray_image`, `near_cam.gray_image`;
    cfg = multiview_cfg(viewpoint_cam, near_cam, W, H, patch_size=config.patch_size, pixel_noise_threshold=config.pixel_noise_threshold).
    `indices` (int32 pixel indices, -1 = unused) overrides the random sample of at most `num_sample` (= config.nunm_sample) valid pixels.
    Both losses are 0 with zero gradients when their mask is empty (the reference's `if d_mask.sum() > 0` / `if mask.sum() > 0`), decided on
    the device: no host synchronisation.
    `out_all_map=` (the rasterizer's (5,H,W) output) may replace `rendered_normal` / `rendered_distance` (pass None for both): the loss then sends
    ONE gradient tensor to out_all_map instead of two sliced ones that autograd pads to five channels and adds up."""
    out = _PlaneMultiview.apply(plane_depth, near_plane_depth, rendered_normal, rendered_distance, gray, near_gray, cfg, lambda_geo, lambda_ncc,
                                int(num_sample), indices, generator, out_all_map)
    if return_aux:
        return out[0], out[1], dict(zip(_MV_AUX, out[2:]))
    return out[0], out[1]


def plane_losses(plane_depth, near_plane_depth, out_all_map, gray, near_gray, cfg, ray_mat, weight=None, lambda_normal=0.015, lambda_geo=0.03,
                 lambda_ncc=0.15, num_sample=102400, indices=None, generator=None):
    """The three PGSR geometry losses of one view after step 7000 -- plane_geo_loss (single-view normal, pgsr_scene.py:105-112) and the two of
    plane_multiview_loss (:113-204) -- as ONE autograd node: -> (normal_loss, geo_loss, ncc_loss), the same values and the same total gradient as the
    two separate calls.  plane_depth and out_all_map feed all three; evaluated separately, autograd sums their gradient maps with two image-sized add
    launches per iteration (5 x H x W and H x W), here they leave backward already summed (gsr_loss_plane_mv_scale with addends)."""
    out = _PlaneMultiview.apply(plane_depth, near_plane_depth, None, None, gray, near_gray, cfg, lambda_geo, lambda_ncc, int(num_sample), indices, generator,
                                out_all_map, (ray_mat, weight, float(lambda_normal)))
    return out[2], out[0], out[1]
