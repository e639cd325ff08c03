"""Per-iteration densification statistics of the explicit-Gaussian methods (include/gsrast.h gsr_densify_stats).

Replaces the statistics half of `VanillaGaussian.densify` / `PGSRGaussian.densify` (gssr/gaussian/vanilla_gaussian.py:467-472,428-430;
gssr/gaussian/pgsr_gaussian.py:164-172,157-161): three (PGSR: five) boolean-mask index assignments, each a nonzero() host synchronisation,
become one elementwise launch."""
import torch

from . import check, dev_f32, lib, ptr, stream_ptr


def _acc(t, name, P):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == P):
        raise RuntimeError(f"densification_stats_: {name} must be a contiguous float32 HIP tensor with one entry per Gaussian")
    return t


def _u8(t):
    t = t.reshape(-1)
    return t.contiguous().view(torch.uint8) if t.dtype == torch.bool else t.to(torch.uint8).contiguous()


def densification_stats_(max_radii2D, xyz_gradient_accum, denom, viewspace_grad, visibility_filter, radii, out_observe=None,
                         viewspace_grad_abs=None, xyz_gradient_accum_abs=None, denom_abs=None):
    """In place.  3DGS / 2DGS: (max_radii2D, xyz_gradient_accum, denom, viewspace_points.grad, visibility_filter, radii).
    PGSR: additionally out_observe, viewspace_points_abs.grad, xyz_gradient_accum_abs, denom_abs."""
    with torch.no_grad():
        g = dev_f32(viewspace_grad, "viewspace_grad")
        P = g.shape[0]
        if P == 0:
            return
        ga = None
        if viewspace_grad_abs is not None:
            ga = dev_f32(viewspace_grad_abs, "viewspace_grad_abs")
            if ga.shape != g.shape or xyz_gradient_accum_abs is None or denom_abs is None:
                raise RuntimeError("densification_stats_: the abs gradient needs its accumulators and the same shape as viewspace_grad")
            _acc(xyz_gradient_accum_abs, "xyz_gradient_accum_abs", P); _acc(denom_abs, "denom_abs", P)
        _acc(max_radii2D, "max_radii2D", P); _acc(xyz_gradient_accum, "xyz_gradient_accum", P); _acc(denom, "denom", P)
        f = _u8(visibility_filter)
        r = radii.reshape(-1).to(torch.int32).contiguous()
        ob = None if out_observe is None else out_observe.reshape(-1).to(torch.int32).contiguous()
        if f.numel() != P or r.numel() != P or (ob is not None and ob.numel() != P):
            raise RuntimeError("densification_stats_: visibility_filter / radii / out_observe must have one entry per Gaussian")
        check(lib().gsr_densify_stats(P, ptr(f), ptr(r), ptr(ob), ptr(g), int(g.shape[1]), ptr(ga), ptr(max_radii2D), ptr(xyz_gradient_accum),
                                      ptr(denom), ptr(xyz_gradient_accum_abs), ptr(denom_abs), stream_ptr(g.device)), "densify_stats")
