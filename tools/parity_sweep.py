"""Randomised parity sweep (GPU box): HIP rasterizer vs the CPU oracle over random regimes (splat size 0.8-40 px, both poses, scale modifier,
near-gate opacities, needle-shaped splats, small odd image sizes), with the NOISE FLOOR measured beside it: the same oracle compiled with
FMA contraction (-ffp-contract=fast) against itself.  A case whose HIP error exceeds the tolerance is only a finding when it also exceeds
that floor -- discontinuous gates (alpha < 1/255, T < 1e-4, T > 0.5, rho3d <= rho2d) and ill-conditioned plane depths make some regimes
differ by 1e-2 between two correct fp32 evaluation orders.  Prints one JSON summary line."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hiprun      # noqa: E402
import oracle      # noqa: E402
import scenes      # noqa: E402

KEYS = {"dL_dmeans3D": "dL_dmeans3D", "dL_dscales": "dL_dscales", "dL_drotations": "dL_drotations", "dL_dopacities": "dL_dopacity"}


def oracle_run(lib_path, sc, variant, og):
    orig = oracle.build
    oracle._LIB = None
    if lib_path:
        oracle.build = lambda force=False: lib_path
    try:
        with oracle.Forward(sc, variant) as f:
            return f.backward(**og), f.color.copy(), f.radii.copy()
    finally:
        oracle.build = orig
        oracle._LIB = None


def rel(a, b):
    return float(np.linalg.norm(a - b.reshape(a.shape)) / (np.linalg.norm(b) + 1e-30))


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 36
    fma = "/tmp/liboracle_fma.so"
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("gsr_oracle.c", "gsd_oracle.c", "gsl_oracle.c")]
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-fopenmp", "-ffp-contract=fast", "-march=native", "-shared", "-o", fma] + srcs + ["-lm"])
    rng = np.random.default_rng(1)
    runs = over_tol = findings = 0
    worst = []
    for seed in range(n_seeds):
        for variant in ("surfel", "ewa", "plane"):
            kw = dict(sigma_px=float(np.exp(rng.uniform(np.log(0.8), np.log(40)))), pose=int(rng.integers(0, 2)),
                      scale_modifier=float(rng.uniform(0.5, 2.0)))
            W, H, P = int(rng.integers(40, 300)), int(rng.integers(40, 220)), int(rng.integers(50, 4000))
            o1 = rng.uniform(0.003, 0.02) if seed % 5 == 0 else None
            sc = scenes.make_scene(variant, P, W, H, seed=2000 + seed, **kw)
            if o1 is not None:
                sc["opacities"][:] = o1
            if seed % 7 == 0:
                sc["scales"][:, 0] *= 20.0
            og = scenes.random_out_grads(variant, W, H, seed=seed, scale=1.0)
            res = hiprun.run(variant, sc, og, device="cuda:0")
            g, color, radii = oracle_run(None, sc, variant, og)
            g2, color2, _ = oracle_run(fma, sc, variant, og)
            runs += 1
            assert np.array_equal(res["radii"], radii), (variant, seed)           # integer outputs are always bit-exact
            e_img = float((np.abs(res["color"] - color) > 1e-4).mean()); f_img = float((np.abs(color2 - color) > 1e-4).mean())
            e_g = max(rel(res["grads"][k], g[kk]) for k, kk in KEYS.items()); f_g = max(rel(g2[kk], g[kk]) for kk in KEYS.values())
            if e_img > 1e-4 or e_g > 1e-3:
                over_tol += 1
                if e_img > 3 * f_img + 1e-4 or e_g > 3 * f_g + 1e-3:
                    findings += 1
                worst.append({"variant": variant, "seed": seed, "hip_grad_relL2": round(e_g, 5), "floor_grad_relL2": round(f_g, 5),
                              "hip_px_frac": round(e_img, 6), "floor_px_frac": round(f_img, 6)})
    print(json.dumps({"runs": runs, "radii_bit_exact": runs, "over_nominal_tolerance": over_tol, "beyond_noise_floor": findings, "cases": worst}))


if __name__ == "__main__":
    main()
