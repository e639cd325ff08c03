"""ctypes binding of the neural-Gaussian decode oracle (oracle/gsd_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

import oracle

_fp = C.POINTER(C.c_float)
PARAM_NAMES = ("W1o", "b1o", "W2o", "b2o", "W1c", "b1c", "W2c", "b2c", "W1k", "b1k", "W2k", "b2k", "app")


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("Na", "Nv", "k", "A", "dist_o", "dist_c", "dist_k", "level")]


class Params(C.Structure):
    _fields_ = [(n, _fp) for n in PARAM_NAMES]


class Inputs(C.Structure):
    _fields_ = [("anchor", _fp), ("feat", _fp), ("offset", _fp), ("scaling", _fp), ("level", _fp), ("opacity_scale", _fp),
                ("vis_idx", C.POINTER(C.c_int32)), ("campos", _fp)]


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


def _pack(case):
    """case: dict from decode_cases.make_case -> (cfg, inputs struct, params struct, keep-alive list)"""
    k = case["k"]
    arr = {n: _f(case[n]) for n in ("anchor", "feat", "offset", "scaling", "level", "opacity_scale", "campos")}
    vis = np.ascontiguousarray(case["vis_idx"], dtype=np.int32)
    par = {n: _f(case["params"].get(n)) for n in PARAM_NAMES}
    A = 0 if par["app"] is None else par["app"].size
    cfg = Cfg(arr["anchor"].shape[0], vis.size, k, A, int(case["dist_o"]), int(case["dist_c"]), int(case["dist_k"]),
              int(arr["level"] is not None))
    inp = Inputs(_p(arr["anchor"]), _p(arr["feat"]), _p(arr["offset"]), _p(arr["scaling"]), _p(arr["level"]), _p(arr["opacity_scale"]),
                 vis.ctypes.data_as(C.POINTER(C.c_int32)), _p(arr["campos"]))
    prm = Params(*[_p(par[n]) for n in PARAM_NAMES])
    return cfg, inp, prm, (arr, vis, par)


# ---- use_feat_bank branch (scaffold_scene.py:45-56): numpy restatement in front of the C oracle, float64 inside
def _bank(case):
    p = case["params"]
    vis = np.asarray(case["vis_idx"], np.int64)
    a = case["anchor"][vis].astype(np.float64); f = case["feat"][vis].astype(np.float64)
    ob = a - case["campos"].astype(np.float64)
    dist = np.linalg.norm(ob, axis=1, keepdims=True); view = ob / dist
    x = np.concatenate([view, dist], 1)
    pre = x @ p["W1b"].astype(np.float64).T + p["b1b"]; h = np.maximum(pre, 0)
    z = h @ p["W2b"].astype(np.float64).T + p["b2b"]
    e = np.exp(z - z.max(1, keepdims=True)); w = e / e.sum(1, keepdims=True)
    f4 = np.tile(f[:, ::4], (1, 4)); f2 = np.tile(f[:, ::2], (1, 2))
    fb = f4 * w[:, :1] + f2 * w[:, 1:2] + f * w[:, 2:]
    return dict(vis=vis, ob=ob, dist=dist, view=view, x=x, pre=pre, h=h, w=w, f=f, f4=f4, f2=f2, fb=fb)


def _with_bank(case):
    """-> (case for the C oracle with the blended features and without the bank parameters, cache)"""
    if case["params"].get("W1b") is None:
        return case, None
    c = _bank(case)
    feat = np.zeros_like(case["feat"]); feat[c["vis"]] = c["fb"].astype(np.float32)
    inner = dict(case); inner["feat"] = feat
    inner["params"] = {k: v for k, v in case["params"].items() if not k.endswith("b") or k in ("b1o", "b2o", "b1c", "b2c", "b1k", "b2k")}
    return inner, c


def _bank_backward(case, c, out):
    """chains the C oracle's gradient w.r.t. the blended features (rows of the visible anchors) through the blend"""
    p = case["params"]
    gfb = out["feat"][c["vis"]].astype(np.float64)
    w, f, f4, f2 = c["w"], c["f"], c["f4"], c["f2"]
    gw = np.stack([(gfb * f4).sum(1), (gfb * f2).sum(1), (gfb * f).sum(1)], 1)
    gf = gfb * w[:, 2:]
    t4 = gfb * w[:, :1]; t2 = gfb * w[:, 1:2]
    for j in range(32):
        gf[:, (j % 8) * 4] += t4[:, j]
        gf[:, (j % 16) * 2] += t2[:, j]
    gz = w * (gw - (gw * w).sum(1, keepdims=True))
    gW2 = gz.T @ c["h"]; gb2 = gz.sum(0)
    gh = (gz @ p["W2b"].astype(np.float64)) * (c["pre"] > 0)
    gW1 = gh.T @ c["x"]; gb1 = gh.sum(0)
    gx = gh @ p["W1b"].astype(np.float64)
    gview, gdist = gx[:, :3], gx[:, 3:4]
    gob = gview / c["dist"] - (gview * c["view"]).sum(1, keepdims=True) * c["view"] / c["dist"] + gdist * c["view"]
    feat = np.zeros_like(out["feat"]); feat[c["vis"]] = gf.astype(np.float32)
    out["feat"] = feat
    out["anchor"] = out["anchor"].copy(); out["anchor"][c["vis"]] += gob.astype(np.float32)
    out.update(W1b=gW1.astype(np.float32), b1b=gb1.astype(np.float32), W2b=gW2.astype(np.float32), b2b=gb2.astype(np.float32))
    return out


def forward(case):
    case, _ = _with_bank(case)
    L = oracle.lib()
    L.refd_forward.restype = C.c_int64
    cfg, inp, prm, keep = _pack(case)
    n = cfg.Nv * cfg.k
    out = {"neural_opacity": np.zeros(n, np.float32), "mask": np.zeros(n, np.uint8), "xyz": np.zeros((n, 3), np.float32),
           "color": np.zeros((n, 3), np.float32), "opacity": np.zeros(n, np.float32), "scaling": np.zeros((n, 3), np.float32),
           "rot": np.zeros((n, 4), np.float32)}
    P = L.refd_forward(C.byref(cfg), C.byref(inp), C.byref(prm), _p(out["neural_opacity"]),
                       out["mask"].ctypes.data_as(C.POINTER(C.c_uint8)), _p(out["xyz"]), _p(out["color"]), _p(out["opacity"]),
                       _p(out["scaling"]), _p(out["rot"]))
    for n_ in ("xyz", "color", "opacity", "scaling", "rot"):
        out[n_] = out[n_][:P].copy()
    out["P"] = int(P)
    return out


def backward(case, mask, dL):
    """dL: dict xyz/color/opacity/scaling/rot (compacted rows) -> dict of gradients (anchor, feat, offset, scaling, params...)."""
    outer = case
    case, bank = _with_bank(case)
    L = oracle.lib()
    L.refd_backward.restype = None
    cfg, inp, prm, keep = _pack(case)
    par = keep[2]
    g = {n: (None if par[n] is None else np.zeros_like(par[n])) for n in PARAM_NAMES}
    gp = Params(*[_p(g[n]) for n in PARAM_NAMES])
    Na, k = cfg.Na, cfg.k
    out = {"anchor": np.zeros((Na, 3), np.float32), "feat": np.zeros((Na, 32), np.float32), "offset": np.zeros((Na, k, 3), np.float32),
           "scaling": np.zeros((Na, 6), np.float32)}
    d = {n: _f(dL[n]) for n in ("xyz", "color", "opacity", "scaling", "rot")}
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    L.refd_backward(C.byref(cfg), C.byref(inp), C.byref(prm), m.ctypes.data_as(C.POINTER(C.c_uint8)), _p(d["xyz"]), _p(d["color"]),
                    _p(d["opacity"]), _p(d["scaling"]), _p(d["rot"]), _p(out["anchor"]), _p(out["feat"]), _p(out["offset"]),
                    _p(out["scaling"]), C.byref(gp))
    out.update({n: g[n] for n in PARAM_NAMES if g[n] is not None})
    if bank is not None:
        out = _bank_backward(outer, bank, out)
    return out


def lod_mask(anchor, level, extra_level, campos, voxel_size, fork, standard_dist, resolution_scale, coarse_index, mode):
    L = oracle.lib()
    L.refd_lod_mask.restype = None
    a = _f(anchor); lv = np.ascontiguousarray(level, dtype=np.int32); ex = _f(extra_level); cp = _f(campos)
    Na = a.shape[0]
    m = np.zeros(Na, np.uint8); pr = np.zeros(Na, np.float32); tr = np.zeros(Na, np.uint8)
    u8 = C.POINTER(C.c_uint8)
    L.refd_lod_mask(C.c_int32(Na), _p(a), lv.ctypes.data_as(C.POINTER(C.c_int32)), _p(ex), _p(cp), C.c_float(voxel_size), C.c_float(fork),
                    C.c_float(standard_dist), C.c_float(resolution_scale), C.c_int32(coarse_index), C.c_int32(mode), m.ctypes.data_as(u8), _p(pr),
                    tr.ctypes.data_as(u8))
    return m.astype(bool), pr, tr.astype(bool)


def training_stats(vis_idx, k, neural_opacity, mask, update_filter, grad, opacity_accum, anchor_demon, offset_gradient_accum, offset_denom):
    """In place on the four float32 accumulators (oracle/gsd_oracle.c refd_training_stats)."""
    L = oracle.lib()
    L.refd_training_stats.restype = None
    vi = np.ascontiguousarray(vis_idx, dtype=np.int32); no = _f(neural_opacity).reshape(-1)
    m = np.ascontiguousarray(mask, dtype=np.uint8); u = np.ascontiguousarray(update_filter, dtype=np.uint8); g = _f(grad)
    u8 = C.POINTER(C.c_uint8)
    for a in (opacity_accum, anchor_demon, offset_gradient_accum, offset_denom):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    L.refd_training_stats(C.c_int32(vi.size), C.c_int32(k), vi.ctypes.data_as(C.POINTER(C.c_int32)), _p(no), m.ctypes.data_as(u8), u.ctypes.data_as(u8),
                          _p(g), C.c_int32(g.shape[1]), _p(opacity_accum), _p(anchor_demon), _p(offset_gradient_accum), _p(offset_denom))
