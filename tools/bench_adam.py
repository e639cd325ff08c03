"""Times one optimizer step over n float32 parameters: torch.optim.Adam(fused=True), torch.optim.Adam(foreach=False) -- what the reference
constructs -- and gsrast.optim.Adam (gsr_adam_step).  28 bytes per parameter are moved."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd"))
from gsrast.optim import Adam   # noqa: E402

dev = torch.device("cuda:0")
out = {}
for n in (3_900_000, 17_700_000):
    res = {}
    for name, mk in (("torch_fused", lambda p: torch.optim.Adam([p], lr=1e-3, eps=1e-15, fused=True)),
                     ("torch_single_tensor", lambda p: torch.optim.Adam([p], lr=1e-3, eps=1e-15, foreach=False)),
                     ("gsrast", lambda p: Adam([p], lr=1e-3, eps=1e-15))):
        p = torch.nn.Parameter(torch.randn(n, device=dev)); p.grad = torch.randn(n, device=dev)
        opt = mk(p)
        for _ in range(5):
            opt.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            opt.step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        res[name] = {"ms": round(ms, 4), "GBps": round(28.0 * n / (ms * 1e-3) / 1e9, 1)}
    out[str(n)] = res
print(json.dumps(out))
