"""Seeded synthetic inputs for the neural-Gaussian decode (anchors, features, offsets, MLP weights)."""
import numpy as np


def make_case(Na=500, k=10, A=32, dist_o=False, dist_c=False, dist_k=False, level=False, progressive=False, vis_frac=0.7, seed=0, feat_bank=False):
    r = np.random.default_rng(seed)
    F = 32
    c = {"k": k, "dist_o": dist_o, "dist_c": dist_c, "dist_k": dist_k}
    c["anchor"] = r.uniform(-5, 5, (Na, 3)).astype(np.float32)
    c["feat"] = r.normal(0, 1, (Na, F)).astype(np.float32)
    c["offset"] = r.normal(0, 0.5, (Na, k, 3)).astype(np.float32)
    c["scaling"] = np.exp(r.normal(-2, 0.5, (Na, 6))).astype(np.float32)
    c["level"] = r.integers(0, 5, Na).astype(np.float32) if level else None
    c["opacity_scale"] = r.uniform(0.2, 1.0, Na).astype(np.float32) if progressive else None
    c["campos"] = np.array([0.3, -0.2, 7.5], np.float32)
    vis = np.nonzero(r.uniform(size=Na) < vis_frac)[0].astype(np.int32)
    c["vis_idx"] = vis
    lv = 1 if level else 0

    def lin(o, i):
        b = 1.0 / np.sqrt(i)
        return r.uniform(-b, b, (o, i)).astype(np.float32), r.uniform(-b, b, o).astype(np.float32)
    p = {}
    p["W1o"], p["b1o"] = lin(32, 35 + int(dist_o) + lv); p["W2o"], p["b2o"] = lin(k, 32)
    p["W1c"], p["b1c"] = lin(32, 35 + int(dist_c) + lv); p["W2c"], p["b2c"] = lin(7 * k, 32)
    p["W1k"], p["b1k"] = lin(32, 35 + int(dist_k) + lv + A); p["W2k"], p["b2k"] = lin(3 * k, 32)
    p["app"] = r.normal(0, 1, A).astype(np.float32) if A else None
    if feat_bank:      # mlp_feature_bank: Linear(view_dim + 1 = 4, 32) - ReLU - Linear(32, 3) - Softmax (scaffold_gaussian.py:133-139)
        p["W1b"], p["b1b"] = lin(32, 4); p["W2b"], p["b2b"] = lin(3, 32)
    c["params"] = p
    return c


def make_out_grads(P, seed=0):
    r = np.random.default_rng(1000 + seed)
    return {"xyz": r.normal(0, 1, (P, 3)).astype(np.float32), "color": r.normal(0, 1, (P, 3)).astype(np.float32),
            "opacity": r.normal(0, 1, P).astype(np.float32), "scaling": r.normal(0, 1, (P, 3)).astype(np.float32),
            "rot": r.normal(0, 1, (P, 4)).astype(np.float32)}
