"""Kernel time of the fused image-space losses at 1080p, C ABI called back to back on preallocated buffers (no autograd, no allocation):
gsr_loss_l1_ssim (k_ssim_fwd + k_ssim_bwd + finish), gsr_loss_surfel_geo, gsr_loss_plane_geo.  One JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd"))
from gsrast import lib, ptr, stream_ptr   # noqa: E402

H, W = 1080, 1920
dev = torch.device("cuda:0")
L = lib()
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.rand(s, generator=g).to(dev)
sp = stream_ptr(dev)


def timed(fn, n=100, w=10):
    for _ in range(w):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


out = {}
img, gt = rnd(3, H, W), rnd(3, H, W)
o3 = torch.empty(3, device=dev); dimg = torch.empty_like(img)
scr = torch.empty(L.gsr_loss_l1_ssim_scratch_bytes(3, H, W), dtype=torch.uint8, device=dev)
out["l1_ssim_ms"] = timed(lambda: L.gsr_loss_l1_ssim(3, H, W, ptr(img), ptr(gt), 0.2, ptr(o3), ptr(dimg), ptr(scr), scr.numel(), sp))
out["l1_ssim_value"] = [round(float(v), 6) for v in o3]
out["l1_ssim_grad_abs_sum"] = round(float(dimg.abs().sum()), 6)

am = rnd(11, H, W); am[0] += 1.0; am[1] = am[1] * 0.5 + 0.5; am[5] += 1.0
rm = torch.tensor([[1e-3, 0, 0], [0, 1e-3, 0], [-0.96, -0.54, 1]], device=dev); nr = torch.eye(3, device=dev)
dL = torch.empty_like(am)
scr2 = torch.empty(max(L.gsr_loss_surfel_geo_scratch_bytes(H, W), 8), dtype=torch.uint8, device=dev)
out["surfel_geo_ms"] = timed(lambda: L.gsr_loss_surfel_geo(H, W, ptr(am), ptr(rm), ptr(nr), 0.0, 0.05, 100.0, ptr(o3), ptr(dL), None, None, None,
                                                           ptr(scr2), scr2.numel(), sp))
out["surfel_geo_value"] = [round(float(v), 6) for v in o3]
out["surfel_geo_grad_abs_sum"] = round(float(dL.abs().sum()), 6)

depth, alpha, nrm, wgt = rnd(1, H, W) + 1.0, rnd(1, H, W), rnd(3, H, W) - 0.5, rnd(1, H, W)
dD, dN = torch.empty_like(depth), torch.empty_like(nrm)
out["plane_geo_ms"] = timed(lambda: L.gsr_loss_plane_geo(H, W, ptr(depth), ptr(alpha), ptr(nrm), ptr(wgt), ptr(rm), 0.015, ptr(o3), ptr(dD), ptr(dN),
                                                         None, ptr(scr2), scr2.numel(), sp))
out["plane_geo_value"] = [round(float(v), 6) for v in o3]
out["plane_geo_grad_abs_sum"] = round(float(dD.abs().sum() + dN.abs().sum()), 6)
print(json.dumps(out))
