"""Times the fused L1+SSIM loss (gsrast.losses.l1_ssim: k_ssim_fwd + k_ssim_bwd + k_ssim_finish) at 3 x 1080 x 1920: ms per call, value + gradient."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd"))
from gsrast.losses import l1_ssim   # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
img = torch.rand((3, 1080, 1920), generator=g).to(dev).requires_grad_(True)
gt = torch.rand((3, 1080, 1920), generator=g).to(dev)
for _ in range(5):
    l1_ssim(img, gt, 0.2).backward(); img.grad = None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50):
    loss = l1_ssim(img, gt, 0.2); loss.backward(); img.grad = None
e1.record(); torch.cuda.synchronize()
print(json.dumps({"l1_ssim_fwd_bwd_ms": round(e0.elapsed_time(e1) / 50, 4), "loss": float(loss)}))
