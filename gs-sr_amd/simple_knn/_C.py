"""Drop-in replacement for `simple_knn._C` (submodules/simple-knn/ext.cpp:15-17):
    distCUDA2(points float32 [P,3]) -> float32 [P]   mean squared distance to the 3 nearest neighbours."""
import torch

from gsrast import lib, check, ptr, stream_ptr, dev_f32


def distCUDA2(points):
    pts = dev_f32(points, "points", allow_empty=True)
    P = int(points.size(0))
    out = torch.zeros((P,), dtype=torch.float32, device=points.device)
    if P == 0:
        return out
    L = lib()
    scratch = torch.empty((max(int(L.gsr_dist2_scratch_bytes(P)), 1),), dtype=torch.uint8, device=points.device)
    check(L.gsr_dist2(P, ptr(pts), ptr(out), ptr(scratch), scratch.numel(), stream_ptr(points.device)), "distCUDA2")
    return out
