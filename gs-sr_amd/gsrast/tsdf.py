"""HIP TSDF voxel integration: the per-frame update of GaussianExtractor.extract_mesh_unbounded's
compute_unbounded_tsdf (gssr/utils/mesh_utils.py:195-246) as one streaming kernel."""
import torch

from . import lib, check, ptr, stream_ptr, dev_f32


def tsdf_integrate_(points, full_proj_transform, depthmap, rgbmap, sdf_trunc, tsdfs, rgbs, weights):
    """In place.  points [V,3]; depthmap [1,H,W] or [H,W]; rgbmap [3,H,W]; sdf_trunc float or tensor [V];
    tsdfs [V], rgbs [V,3], weights [V] float32 CUDA tensors (initial values: 1, 0, 1 as in the reference)."""
    pts = dev_f32(points, "points", allow_empty=False)
    F = dev_f32(full_proj_transform, "full_proj_transform", allow_empty=False)
    d = dev_f32(depthmap, "depthmap", allow_empty=False)
    c = dev_f32(rgbmap, "rgbmap", allow_empty=False)
    H, W = int(d.shape[-2]), int(d.shape[-1])
    for t, n in ((tsdfs, "tsdfs"), (rgbs, "rgbs"), (weights, "weights")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"{n} must be a contiguous float32 CUDA tensor")
    tp = None
    st = 0.0
    if isinstance(sdf_trunc, torch.Tensor):
        tp = dev_f32(sdf_trunc, "sdf_trunc", allow_empty=False)
    else:
        st = float(sdf_trunc)
    check(lib().gsr_tsdf_integrate(int(pts.shape[0]), ptr(pts), ptr(F), W, H, ptr(d), ptr(c), st, ptr(tp), ptr(tsdfs),
                                   ptr(weights), ptr(rgbs), stream_ptr(pts.device)), "tsdf_integrate")
    return tsdfs, rgbs, weights
