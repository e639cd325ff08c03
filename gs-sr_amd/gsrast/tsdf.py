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
    scratch = torch.empty((H * W, 4), dtype=torch.float32, device=pts.device)      # interleaved (r,g,b,d) texels
    check(lib().gsr_tsdf_integrate(int(pts.shape[0]), ptr(pts), ptr(F), W, H, ptr(d), ptr(c), st, ptr(tp), ptr(tsdfs),
                                   ptr(weights), ptr(rgbs), ptr(scratch), stream_ptr(pts.device)), "tsdf_integrate")
    return tsdfs, rgbs, weights


class DenseTSDFVolume:
    """Dense-grid stand-in for o3d.pipelines.integration.ScalableTSDFVolume as GS-SR uses it
    (gssr/utils/mesh_utils.py:154-178: voxel_length, sdf_trunc, RGB8 colours, depth_scale=1, depth_trunc).
    Open3D is not vendored in the reference: the voxel update follows Open3D's published UniformTSDFVolume algorithm and
    its parity is UNPINNED.  State: tsdf, weight [nx,ny,nz], color [nx,ny,nz,3] float32 on the HIP device.

    Multi-GPU (extract_mesh_split.py fuses the frames of all tiles into one volume): every rank integrates its own
    frames into its own volume, then merge_() all-reduces the two associative accumulators (sum w*tsdf, sum w) over
    RCCL -- the weighted average is order-independent up to fp32 rounding."""

    def __init__(self, origin, voxel_length, dims, sdf_trunc, device="cuda"):
        self.origin = [float(v) for v in origin]
        self.voxel_length = float(voxel_length)
        self.dims = tuple(int(d) for d in dims)
        self.sdf_trunc = float(sdf_trunc)
        self.device = torch.device(device)
        self.tsdf = torch.zeros(self.dims, dtype=torch.float32, device=self.device)
        self.weight = torch.zeros(self.dims, dtype=torch.float32, device=self.device)
        self.color = torch.zeros(self.dims + (3,), dtype=torch.float32, device=self.device)

    def integrate(self, rgb, depth, fx, fy, cx, cy, extrinsic, depth_trunc=float("inf"), quantize_rgb8=True):
        """rgb [3,H,W] in [0,1], depth [1,H,W] or [H,W], extrinsic 4x4 world->camera (row-major, Open3D convention)."""
        import ctypes as C
        d = dev_f32(depth, "depth", allow_empty=False)
        c = dev_f32(rgb, "rgb", allow_empty=False)
        if quantize_rgb8:        # mesh_utils.py:170 converts colours to uint8 before fusion
            c = (torch.clamp(c, 0.0, 1.0) * 255).to(torch.uint8).to(torch.float32).contiguous()
        H, W = int(d.shape[-2]), int(d.shape[-1])
        E = (C.c_float * 16)(*[float(v) for v in torch.as_tensor(extrinsic, dtype=torch.float32).reshape(-1).tolist()])
        o = (C.c_float * 3)(*self.origin)
        check(lib().gsr_tsdf_integrate_dense(self.dims[0], self.dims[1], self.dims[2], o, self.voxel_length, self.sdf_trunc,
                                             float(min(depth_trunc, 3.0e38)), W, H, ptr(d), ptr(c), float(fx), float(fy), float(cx),
                                             float(cy), E, ptr(self.tsdf), ptr(self.weight), ptr(self.color),
                                             stream_ptr(self.device)), "tsdf_integrate_dense")
        return self

    def merge_(self, group=None):
        """All-reduce (sum) of the weighted accumulators across ranks; afterwards every rank holds the fused volume."""
        merge_volumes_(self.tsdf, self.weight, self.color, group)
        return self


def merge_volumes_(tsdf, weight, color, group=None):
    """tsdf <- sum_r(w_r*tsdf_r)/sum_r(w_r), color likewise, weight <- sum_r(w_r); in place, any device/backend."""
    import torch.distributed as dist
    acc = tsdf * weight
    cacc = color * weight.unsqueeze(-1)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(cacc, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(weight, op=dist.ReduceOp.SUM, group=group)
    nz = weight > 0
    tsdf.copy_(torch.where(nz, acc / weight.clamp_min(1e-30), torch.zeros_like(acc)))
    color.copy_(torch.where(nz.unsqueeze(-1), cacc / weight.clamp_min(1e-30).unsqueeze(-1), torch.zeros_like(cacc)))
    return tsdf, weight, color
