"""Tiny driver for rocprofv3: 20 iterations of the PGSR multi-view losses at 1080p (value + gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mv_cases
from gsrast.losses import multiview_cfg, plane_multiview_loss
W, H = 1920, 1080
mc = mv_cases.plane_pair(W=W, H=H, seed=11, amp=0.002, tex=25.0)
t = lambda a: torch.tensor(a, device="cuda")
names = ("plane_depth", "near_plane_depth", "rendered_normal", "rendered_distance")
ml = {k: t(mc[k]).requires_grad_(True) for k in names}
g, ng = t(mc["gray"]), t(mc["near_gray"])
cfg = multiview_cfg(mv_cases.cam_ns(mc["view"]), mv_cases.cam_ns(mc["near"]), W, H, near_size=(W, H))
for _ in range(20):
    for v in ml.values(): v.grad = None
    a, b = plane_multiview_loss(*[ml[k] for k in names], g, ng, cfg)
    (a + b).backward()
torch.cuda.synchronize()
print("ok", a.item(), b.item())
