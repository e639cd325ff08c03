"""Activations of the explicit-Gaussian models in front of the rasterizer (include/gsrast.h gsr_gauss_activations[_backward]).

`VanillaGaussianModel.get_scaling / get_rotation / get_opacity` (gssr/gaussian/vanilla_gaussian.py:86-90,250-269) are exp, F.normalize and sigmoid of
three parameter tensors: three torch kernels forward (normalize alone is three) and ~8 in autograd, every iteration of vanilla-3dgs / 2dgs / pgsr.
Here: one streaming HIP kernel each way."""
import torch

from . import check, dev_f32, lib, ptr, stream_ptr


class _GaussActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling_raw, rotation_raw, opacity_raw):
        s = dev_f32(scaling_raw, "scaling", allow_empty=False)
        q = dev_f32(rotation_raw, "rotation", allow_empty=False)
        o = dev_f32(opacity_raw, "opacity", allow_empty=False)
        P = s.shape[0]
        if s.dim() != 2 or not (1 <= s.shape[1] <= 3) or q.shape != (P, 4) or o.numel() != P:
            raise RuntimeError("gaussian_activations: scaling (P, 1..3), rotation (P, 4), opacity (P, 1) expected")
        so, qo, oo = torch.empty_like(s), torch.empty_like(q), torch.empty_like(o)
        check(lib().gsr_gauss_activations(P, int(s.shape[1]), ptr(s), ptr(q), ptr(o), ptr(so), ptr(qo), ptr(oo), stream_ptr(s.device)), "gauss_activations")
        ctx.save_for_backward(so, q, qo, oo)
        ctx.set_materialize_grads(False)
        return so, qo, oo

    @staticmethod
    def backward(ctx, gs, gq, go):
        so, q, qo, oo = ctx.saved_tensors
        P = so.shape[0]
        c = lambda g: None if g is None else g.contiguous()
        gs, gq, go = c(gs), c(gq), c(go)
        ds, dq, do = torch.empty_like(so), torch.empty_like(q), torch.empty_like(oo)
        check(lib().gsr_gauss_activations_backward(P, int(so.shape[1]), ptr(so), ptr(q), ptr(qo), ptr(oo), ptr(gs), ptr(gq), ptr(go), ptr(ds), ptr(dq), ptr(do),
                                                   stream_ptr(so.device)), "gauss_activations_backward")
        return ds, dq, do


def gaussian_activations(scaling_raw, rotation_raw, opacity_raw):
    """-> (exp(scaling_raw), F.normalize(rotation_raw), sigmoid(opacity_raw)), differentiable, one HIP kernel forward and one backward."""
    return _GaussActivations.apply(scaling_raw, rotation_raw, opacity_raw)
