"""BASELINE-size parity report (GPU box): HIP rasterizer vs the CPU oracle at 300k gaussians, 1920x1080 for every variant / seed /
pose / colour mode of tests/test_gpu_parity.py::FULL_CASES.  Prints one JSON object per case: bit-exactness of the integer stages,
fraction of pixels beyond 1e-4 per output channel, gradient errors (relative L2, fraction beyond 1e-3 max, per-element criterion)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hiprun      # noqa: E402
import oracle      # noqa: E402
import scenes      # noqa: E402
from test_gpu_parity import FULL_CASES      # noqa: E402


def frac_bad(a, b, tol=1e-4):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return float((d > tol * max(1.0, float(np.abs(b).max()))).mean()), float(d.max())


def trimmed_l2(a, b, trim=1e-4):
    """relative L2 error after dropping the `trim` fraction of elements with the largest |a - b| (ill-conditioned outliers)"""
    e = np.abs(a - b); k = int(np.ceil(trim * e.size))
    keep = np.argsort(e)[:e.size - k]
    return float(np.linalg.norm((a - b)[keep]) / (np.linalg.norm(b[keep]) + 1e-30))


def grad_stats(a, b, tol=1e-3):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    rms = np.sqrt((b * b).mean())
    return dict(rel_l2=float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)), rel_l2_trim1e4=trimmed_l2(a, b),
                frac_gt_tol_max=float((np.abs(a - b) > tol * (np.abs(b).max() + 1e-30)).mean()),
                frac_per_element=float((np.abs(a - b) > tol * np.abs(b) + tol * rms).mean()))


def main():
    P, W, H = 300000, 1920, 1080
    cases = FULL_CASES if len(sys.argv) < 2 else [c for c in FULL_CASES if c[0] in sys.argv[1:]]
    for variant, cm, seed, pose in cases:
        sc = scenes.make_scene(variant, P, W, H, seed=seed, color_mode=cm, pose=pose, bg=(0.1, 0.3, 0.2) if seed else (0.0, 0.0, 0.0))
        og = scenes.random_out_grads(variant, W, H, seed=seed)
        rep = dict(variant=variant, colour=cm, seed=seed, pose=pose)
        with oracle.Forward(sc, variant) as f:
            g = f.backward(**og)
            st = hiprun.run_raw(variant, sc)
            rep.update(R=int(f.R), radii_equal=bool(np.array_equal(st["radii"], f.radii)),
                       point_list_equal=bool(np.array_equal(st["point_list"], f.point_list())))
            ft, nc = f.image_state()
            rep["n_contrib_mismatch_frac"] = float((st["n_contrib"][0] != nc[0]).mean())
            rep["color"] = frac_bad(st["color"], f.color)
            rep["final_T"] = frac_bad(st["final_T"][0], ft[0])
            if variant == "surfel":
                rep["others"] = {ch: frac_bad(st["others"][ch], f.others[ch]) for ch in range(11)}
                rep["median_idx_mismatch_frac"] = float((st["others"][7] != f.others[7]).mean())
                same = st["others"][7] == f.others[7]
                rep["median_channels_where_idx_agrees"] = {ch: float((np.abs(st["others"][ch] - f.others[ch])[same] > 1e-4 * max(1.0, np.abs(f.others[ch]).max())).mean()) for ch in (5, 8, 9, 10)}
            if variant == "plane":
                rep["all_map"] = frac_bad(st["all_map"], f.out_all_map)
                rep["plane_depth"] = frac_bad(st["plane_depth"], f.plane_depth)
                d = np.abs(st["observe"].astype(np.int64) - f.observe)
                rep["observe"] = dict(max=int(d.max()), n_diff=int((d > 0).sum()))
            res = hiprun.run(variant, sc, og)
        gg = res["grads"]
        pairs = [("dL_dmeans3D", "dL_dmeans3D"), ("dL_dscales", "dL_dscales"), ("dL_drotations", "dL_drotations"),
                 ("dL_dopacities", "dL_dopacity"), ("dL_dmeans2D", "dL_dmeans2D")]
        pairs.append(("dL_dshs", "dL_dsh") if cm == "sh" else ("dL_dcolors_precomp", "dL_dcolors"))
        if variant == "plane":
            pairs += [("dL_dall_map", "dL_dall_map"), ("dL_dmeans2D_abs", "dL_dmeans2D_abs")]
        rep["grads"] = {a: grad_stats(gg[a], g[b]) for a, b in pairs}
        with oracle.fma_twin():
            with oracle.Forward(sc, variant) as f2:
                g2 = f2.backward(**og)
        rep["floor_grads"] = {a: grad_stats(g2[b].reshape(g[b].shape), g[b]) for a, b in pairs}
        print(json.dumps(rep))


if __name__ == "__main__":
    main()
