"""Loaders for tests/golden/ref_*.npz: vectors produced by running the reference's own Python in the authoring container
(tests/golden/make_golden_ref.py).  Data only; nothing here touches /root/reference."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAM_NAMES = ("W1o", "b1o", "W2o", "b2o", "W1c", "b1c", "W2c", "b2c", "W1k", "b1k", "W2k", "b2k", "app", "W1b", "b1b", "W2b", "b2b")
DECODE = ("ref_decode_scaffold", "ref_decode_scaffold_dist", "ref_decode_octree", "ref_decode_scaffold_featbank")


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def decode_case(name):
    """-> (case dict in decode_cases.make_case format, dL dict, expected outputs dict, expected gradients dict)"""
    z = load(name)
    case = {"k": int(z["k"]), "dist_o": bool(z["dist_o"]), "dist_c": bool(z["dist_c"]), "dist_k": bool(z["dist_k"])}
    for n in ("anchor", "feat", "offset", "scaling", "level", "opacity_scale", "campos", "vis_idx"):
        case[n] = z.get("in_" + n)
    case["params"] = {n: z.get("p_" + n) for n in PARAM_NAMES if (z.get("p_" + n) is not None or not n.endswith("b"))}
    dL = {n: z["dL_" + n] for n in ("xyz", "color", "opacity", "scaling", "rot")}
    exp = {n: z[n] for n in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity", "mask")}
    grads = {n[2:]: v for n, v in z.items() if n.startswith("g_")}
    return case, dL, exp, grads
