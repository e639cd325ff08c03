"""BASELINE-size parity report (GPU box): the HIP rasterizer against the FLOAT64 truth at 300k gaussians, 1920x1080 for every variant / seed /
pose / colour mode of tests/test_gpu_parity.py::FULL_CASES, with the float32 oracle's own error against the same truth beside every figure
(tests/parity_truth.py -- the criterion of test_full_size_oracle_parity), then the concentrated scenes of SKEW_CASES (tile lists of thousands of entries) and, with
--big, one P = 1 000 000 case.  One JSON object per case on stdout; EXIT CODE 1 when any case violates a bar (the report gates, it does not only print).
    python tools/full_parity_report.py [--big] [--skew-only] [variant ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hiprun      # noqa: E402
import parity_truth as pt      # noqa: E402
import scenes      # noqa: E402
from test_gpu_parity import FULL_CASES_REPORT as FULL_CASES, SKEW_CASES, _hip_outputs      # noqa: E402


def compact(v):
    if isinstance(v, dict):
        return {k: compact(x) for k, x in v.items()}
    if isinstance(v, (float, np.floating)):
        return float(f"{float(v):.4g}")
    if isinstance(v, np.integer):
        return int(v)
    if isinstance(v, np.bool_):
        return bool(v)
    return v


def main():
    W, H = 1920, 1080
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    cases = [] if "--skew-only" in flags else [dict(variant=v, cm=cm, seed=seed, pose=pose, P=300000) for v, cm, seed, pose in FULL_CASES]
    cases += [dict(variant=v, cm="precomp", seed=0, pose=0, P=300000, skew=(frac, scale)) for v, frac, scale in SKEW_CASES]
    if "--big" in flags:
        cases.append(dict(variant="surfel", cm="precomp", seed=0, pose=0, P=1000000))
    if names:
        cases = [c for c in cases if c["variant"] in names]
    failed = 0
    for c in cases:
        variant, seed, pose, P = c["variant"], c["seed"], c["pose"], c["P"]
        cm, _, deg = c["cm"].partition(":")          # "sh:d" = (P,16,3) coefficients at active degree d
        sc = scenes.make_scene(variant, P, W, H, seed=seed, color_mode=cm, pose=pose, bg=(0.1, 0.3, 0.2) if seed else (0.0, 0.0, 0.0),
                               sh_degree=int(deg or 3))
        if c.get("skew"):
            scenes.concentrate(sc, *c["skew"])
        og = scenes.random_out_grads(variant, W, H, seed=seed)
        st, cand = _hip_outputs(hiprun, variant, sc, og)
        f32, fma, truth, ints = pt.run_oracles(sc, variant, og, hip_state=st)      # incl. the filtered instance list against the oracle's (tests/tile_cull.py)
        cand["n_contrib"] = ints["view"]["n_contrib"]                              # positions in the oracle's list
        rep, verdict = {}, "pass"
        try:
            pt.check_case(variant, cm, cand, f32, fma, truth, rep, min_robust=0.9 if P <= 300000 else 0.8)
        except AssertionError as e:
            verdict = "FAIL: " + str(e)
            failed += 1
        ln = np.diff(st["ranges"].astype(np.int64), axis=1)[:, 0]
        out = dict(case=dict(variant=variant, color_mode=cm, sh_degree=int(deg or 3) if cm == "sh" else None, seed=seed, pose=pose, P=P, W=W, H=H, R=int(st["R"]),
                             min_robust_pixel_fraction=0.9 if P <= 300000 else 0.8, skew=c.get("skew"), tile_list_max=int(ln.max()), tile_lists_over_1024=int((ln > 1024).sum())), verdict=verdict,
                   radii_bit_exact=bool(np.array_equal(st["radii"], ints["radii"])),
                   tile_instances=compact({k: v for k, v in ints["view"].items() if k not in ("keep", "n_contrib")}),
                   report=compact(rep))
        print(json.dumps(out), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
