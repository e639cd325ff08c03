"""Per-Gaussian `all_map` input of the plane rasterizer (include/gsrast.h gsr_plane_allmap[_backward]).

Replaces, inside `PGSRScene.render()` (gssr/scene/pgsr_scene.py:297-304, with get_normal / get_smallest_axis :241-257), the ~25 torch ops
(quaternion_to_matrix, min, gather, masked negate, two matmuls, abs ...) and their ~40 autograd nodes by one streaming kernel each way."""
import torch

from . import check, dev_f32, lib, ptr, stream_ptr


class _PlaneAllMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, rotations, scales, viewmatrix, campos):
        x = dev_f32(means3D, "means3D", allow_empty=False)
        q = dev_f32(rotations, "rotations", allow_empty=False)
        s = dev_f32(scales, "scales", allow_empty=False)
        V = dev_f32(viewmatrix, "viewmatrix", allow_empty=False)
        cp = dev_f32(campos, "campos", allow_empty=False)
        P = x.shape[0]
        if q.shape != (P, 4) or s.dim() != 2 or s.shape[0] != P or s.shape[1] < 3 or V.numel() != 16 or cp.numel() != 3:
            raise RuntimeError("plane_input_all_map: means3D (P,3), rotations (P,4), scales (P,>=3), viewmatrix (4,4), campos (3,) expected")
        out = torch.empty(P, 5, dtype=torch.float32, device=x.device)
        check(lib().gsr_plane_allmap(P, ptr(x), ptr(q), ptr(s), int(s.shape[1]), ptr(V), ptr(cp), ptr(out), stream_ptr(x.device)), "plane_allmap")
        ctx.save_for_backward(x, q, s, V, cp)
        return out

    @staticmethod
    def backward(ctx, g):
        x, q, s, V, cp = ctx.saved_tensors
        g = g.contiguous()
        P = x.shape[0]
        dx = torch.empty_like(x)
        dq = torch.empty_like(q)
        check(lib().gsr_plane_allmap_backward(P, ptr(x), ptr(q), ptr(s), int(s.shape[1]), ptr(V), ptr(cp), ptr(g), ptr(dx), ptr(dq),
                                              stream_ptr(x.device)), "plane_allmap_backward")
        return dx, dq, None, None, None


def plane_input_all_map(means3D, rotations, scales, viewmatrix, campos):
    """-> all_map (P,5) = [local_normal, 1, local_distance], differentiable w.r.t. means3D and rotations exactly as the reference's op chain
    (no gradient to scales: argmin; none through the camera-facing flip).  viewmatrix = viewpoint_camera.world_view_transform,
    campos = viewpoint_camera.camera_center."""
    return _PlaneAllMap.apply(means3D, rotations, scales, viewmatrix, campos)
