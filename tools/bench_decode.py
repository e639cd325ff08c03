"""Times the fused neural-Gaussian decode (gsrast.decode) against the reference's torch op chain on the same GPU.
Workload: Nv visible anchors x k=10 offsets sized so that ~300k Gaussians are emitted (the headline rasterizer workload)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import decode_cases          # noqa: E402
import ref_decode_torch      # noqa: E402
from gsrast import decode    # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    Na = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    case = decode_cases.make_case(Na=Na, seed=0, vis_frac=0.6)
    t = lambda a: None if a is None else torch.tensor(a, device=DEV)
    leaves = {n: t(case[n]).requires_grad_(True) for n in ("anchor", "feat", "offset", "scaling")}
    par = {n: t(v).requires_grad_(True) for n, v in case["params"].items()}
    vis = torch.tensor(case["vis_idx"], dtype=torch.int32, device=DEV)
    campos = t(case["campos"])
    heads = [(par["W1" + h], par["b1" + h], par["W2" + h], par["b2" + h]) for h in "ock"]

    def hip_fwd():
        return decode.neural_gaussians(leaves["anchor"], leaves["feat"], leaves["offset"], leaves["scaling"], *heads, campos, vis_idx=vis,
                                       appearance=par["app"])
    out = hip_fwd()
    P = out[0].shape[0]
    g = [torch.randn_like(o) for o in out[:5]]

    def hip_fwd_bwd():
        o = hip_fwd()
        torch.autograd.backward(o[:5], g)

    o0, _ = ref_decode_torch.decode_live(case, leaves, par, vis.long(), campos)      # its own P: the gate may flip on ~0 values
    gt = [torch.randn_like(o0[n]) for n in ("xyz", "color", "opacity", "scaling", "rot")]

    def torch_chain(bwd):
        o, lv = ref_decode_torch.decode_live(case, leaves, par, vis.long(), campos)
        if bwd:
            torch.autograd.backward([o[n] for n in ("xyz", "color", "opacity", "scaling", "rot")], gt)
    res = {"Na": Na, "Nv": int(vis.numel()), "k": case["k"], "P": int(P),
           "hip_fwd_ms": timeit(hip_fwd), "hip_fwd_bwd_ms": timeit(hip_fwd_bwd),
           "torch_fwd_ms": timeit(lambda: torch_chain(False)), "torch_fwd_bwd_ms": timeit(lambda: torch_chain(True))}
    res["speedup_fwd_bwd"] = res["torch_fwd_bwd_ms"] / res["hip_fwd_bwd_ms"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
