"""ctypes binding of the PGSR multi-view loss oracle (oracle/gsm_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

import oracle

_fp = C.POINTER(C.c_float)


class Cfg(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("W", "H", "Wn", "Hn", "Wg", "Hg")] +
                [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "nfx", "nfy", "ncx", "ncy")] +
                [("v2n", C.c_float * 12), ("n2v", C.c_float * 12), ("ncc_scale", C.c_float), ("noise_th", C.c_float), ("patch", C.c_int32)])


def rigid_pair(Rv, Tv, Rn, Tn):
    """Row-vector cameras X_cam = X_world R + T (gssr/cameras/__init__.py:85: world_view_transform[:3,:3] = R, [3,:3] = T)
    -> (v2n, n2v) as 12 floats each: A row-major then b, X_near = X_view A + b."""
    Rv, Tv, Rn, Tn = (np.asarray(a, np.float64) for a in (Rv, Tv, Rn, Tn))
    A = Rv.T @ Rn; b = Tn - Tv @ A
    Ai = Rn.T @ Rv; bi = Tv - Tn @ Ai
    return (np.concatenate([A.reshape(-1), b]).astype(np.float32), np.concatenate([Ai.reshape(-1), bi]).astype(np.float32))


def make_cfg(W, H, view, near, Wn=None, Hn=None, Wg=None, Hg=None, ncc_scale=1.0, noise_th=1.0, patch=3):
    """view / near: dicts with R, T, Fx, Fy, Cx, Cy."""
    v2n, n2v = rigid_pair(view["R"], view["T"], near["R"], near["T"])
    c = Cfg(W, H, Wn or W, Hn or H, Wg or W, Hg or H, view["Fx"], view["Fy"], view["Cx"], view["Cy"], near["Fx"], near["Fy"], near["Cx"], near["Cy"])
    c.v2n[:] = v2n.tolist(); c.n2v[:] = n2v.tolist()
    c.ncc_scale, c.noise_th, c.patch = ncc_scale, noise_th, patch
    return c


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_fp)


def geo(cfg, depth, near_depth):
    L = oracle.lib()
    L.refm_multiview_geo.restype = None
    d, nd = _f(depth).reshape(-1), _f(near_depth).reshape(-1)
    n = cfg.W * cfg.H
    out = {"noise": np.zeros(n, np.float32), "dmask": np.zeros(n, np.uint8), "weight": np.zeros(n, np.float32), "stats": np.zeros(2, np.float64),
           "g_depth": np.zeros(n, np.float32), "g_near": np.zeros(cfg.Wn * cfg.Hn, np.float32)}
    L.refm_multiview_geo(C.byref(cfg), _p(d), _p(nd), _p(out["noise"]), out["dmask"].ctypes.data_as(C.c_void_p), _p(out["weight"]),
                         out["stats"].ctypes.data_as(C.c_void_p), _p(out["g_depth"]), _p(out["g_near"]))
    return out


def ncc(cfg, idx, weight, normal, dist, gray, near_gray):
    L = oracle.lib()
    L.refm_multiview_ncc.restype = None
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    w, nm, ds, g, ng = _f(weight).reshape(-1), _f(normal).reshape(-1), _f(dist).reshape(-1), _f(gray).reshape(-1), _f(near_gray).reshape(-1)
    n = cfg.W * cfg.H
    out = {"ncc": np.zeros(idx.size, np.float32), "mask": np.zeros(idx.size, np.uint8), "stats": np.zeros(2, np.float64),
           "g_normal": np.zeros((3, n), np.float32), "g_dist": np.zeros(n, np.float32)}
    L.refm_multiview_ncc(C.byref(cfg), C.c_int32(idx.size), idx.ctypes.data_as(C.c_void_p), _p(w), _p(nm), _p(ds), _p(g), _p(ng), _p(out["ncc"]),
                         out["mask"].ctypes.data_as(C.c_void_p), out["stats"].ctypes.data_as(C.c_void_p), _p(out["g_normal"]), _p(out["g_dist"]))
    return out


def fixture_cfg(z):
    cam = lambda pre: {k: (z[f"{pre}_{k}"] if k in ("R", "T") else float(z[f"{pre}_{k}"])) for k in ("R", "T", "Fx", "Fy", "Cx", "Cy")}
    return make_cfg(int(z["W"]), int(z["H"]), cam("v"), cam("n"), noise_th=float(z["pixel_noise_threshold"]), patch=int(z["patch_size"]))


def plane_allmap(xyz, rot, scale, viewmatrix, campos, d_all_map=None):
    """-> all_map [P,5] (and (d_xyz, d_rot) when d_all_map is given): oracle/gsm_oracle.c refm_plane_allmap[_bwd]."""
    L = oracle.lib()
    L.refm_plane_allmap.restype = None; L.refm_plane_allmap_bwd.restype = None
    x, q, s, V, cp = _f(xyz), _f(rot), _f(scale), _f(viewmatrix).reshape(-1), _f(campos)
    P = x.shape[0]
    am = np.zeros((P, 5), np.float32)
    L.refm_plane_allmap(C.c_int32(P), _p(x), _p(q), _p(s), _p(V), _p(cp), _p(am))
    if d_all_map is None:
        return am
    g = _f(d_all_map); dx = np.zeros((P, 3), np.float32); dq = np.zeros((P, 4), np.float32)
    L.refm_plane_allmap_bwd(C.c_int32(P), _p(x), _p(q), _p(s), _p(V), _p(cp), _p(g), _p(dx), _p(dq))
    return am, dx, dq


def densify_stats(filt, radii, grad, max_radii2D, accum, denom, out_observe=None, grad_abs=None, accum_abs=None, denom_abs=None):
    """In place on the float32 accumulators (oracle/gsm_oracle.c refm_densify_stats)."""
    L = oracle.lib()
    L.refm_densify_stats.restype = None
    f = np.ascontiguousarray(filt, dtype=np.uint8); r = np.ascontiguousarray(radii, dtype=np.int32); g = _f(grad)
    ob = None if out_observe is None else np.ascontiguousarray(out_observe, dtype=np.int32)
    ga = None if grad_abs is None else _f(grad_abs)
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    L.refm_densify_stats(C.c_int32(f.size), vp(f), vp(r), vp(ob), _p(g), C.c_int32(g.shape[1]), vp(ga), vp(max_radii2D), vp(accum), vp(denom),
                         vp(accum_abs), vp(denom_abs))
