import json, os, subprocess, sys
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hiprun, oracle, scenes
from parity_sweep import oracle_run
fma = "/tmp/liboracle_fma.so"
srcs = [os.path.join(ROOT, "oracle", f) for f in ("gsr_oracle.c", "gsd_oracle.c", "gsl_oracle.c", "gsm_oracle.c") if os.path.exists(os.path.join(ROOT, "oracle", f))]
subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-fopenmp", "-ffp-contract=fast", "-march=native", "-shared", "-o", fma] + srcs + ["-lm"])
P, W, H = 300000, 1920, 1080
sc = scenes.make_scene("surfel", P, W, H, seed=0)
og = scenes.random_out_grads("surfel", W, H, seed=0)
res = hiprun.run("surfel", sc, og, device="cuda:0")
g, color, radii = oracle_run(None, sc, "surfel", og)
g2, color2, _ = oracle_run(fma, sc, "surfel", og)
for k, kk in (("dL_dmeans3D", "dL_dmeans3D"), ("dL_drotations", "dL_drotations"), ("dL_dscales", "dL_dscales"), ("dL_dmeans2D", "dL_dmeans2D")):
    a = res["grads"][k].astype(np.float64); b = g[kk].astype(np.float64).reshape(a.shape); c = g2[kk].astype(np.float64).reshape(a.shape)
    eh = np.abs(a - b).sum(-1); ef = np.abs(c - b).sum(-1)
    print(k, "relL2 hip", np.linalg.norm(a - b) / np.linalg.norm(b), "floor", np.linalg.norm(c - b) / np.linalg.norm(b), "norm ref", np.linalg.norm(b))
    top = np.argsort(-eh)[:6]
    for t in top:
        print("   gauss", int(t), "ref", b[t], "hip-ref", (a - b)[t], "fma-ref", (c - b)[t])
    # robust: drop the 0.01% largest-|ref| rows
    mag = np.abs(b).sum(-1); keep = mag <= np.quantile(mag, 0.9999)
    print("   relL2 without the top 1e-4 rows by |ref|: hip", np.linalg.norm((a - b)[keep]) / np.linalg.norm(b[keep]), "floor", np.linalg.norm((c - b)[keep]) / np.linalg.norm(b[keep]))
print("color frac>1e-4 hip", float((np.abs(res["color"] - color) > 1e-4).mean()), "floor", float((np.abs(color2 - color) > 1e-4).mean()))
