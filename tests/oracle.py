"""ctypes binding of the CPU oracle (oracle/libgsr_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
(gs-sr_amd/) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

EWA, SURFEL, PLANE = 0, 1, 2
VARIANT_ID = {"ewa": EWA, "surfel": SURFEL, "plane": PLANE}

_fp = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
GATE_NAMES = {0: "none", 1: "power>0", 2: "alpha<1/255", 3: "T(1-alpha)<1e-4", 4: "T>0.5", 5: "rho3d<=rho2d", 6: "depth<near"}     # gsr_oracle.h REF_GATE_*


def _structs(ct):
    """ctypes mirrors of gsr_oracle.h for the arithmetic type `real` = ct (c_float: libgsr_oracle.so / _fma.so, c_double: _f64.so)."""
    rp = C.POINTER(ct)

    class RefInputs(C.Structure):
        _fields_ = [("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
                    ("tanfovx", ct), ("tanfovy", ct), ("scale_modifier", ct),
                    ("prefiltered", C.c_int32), ("render_geo", C.c_int32),
                    ("bg", rp), ("viewmatrix", rp), ("projmatrix", rp), ("campos", rp), ("means3D", rp),
                    ("shs", rp), ("colors_precomp", rp), ("opacities", rp), ("scales", rp), ("rotations", rp),
                    ("cov3D_precomp", rp), ("all_map", rp),
                    ("ov_radii", _i32p), ("ov_point_list", _u32p), ("ov_ranges", _u32p), ("ov_R", C.c_int32),
                    ("gate_margin", rp), ("gate_id", _i32p), ("gate_splat", _i32p), ("splat_noise", rp),
                    ("ov_cov", rp), ("ov_conic_opacity", rp), ("ov_means2D", rp), ("ov_depths", rp), ("ov_rgb", rp)]

    class RefOutGrads(C.Structure):
        _fields_ = [("dL_dcolor", rp), ("dL_dothers", rp), ("dL_dout_all_map", rp), ("dL_dplane_depth", rp)]

    class RefInGrads(C.Structure):
        _fields_ = [(n, rp) for n in ("dL_dmeans3D", "dL_dmeans2D", "dL_dmeans2D_abs", "dL_dcolors", "dL_dopacity",
                                      "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dall_map", "dL_dconic")]
    return RefInputs, RefOutGrads, RefInGrads


RefInputs, RefOutGrads, RefInGrads = _structs(C.c_float)
RefInputs64, RefOutGrads64, RefInGrads64 = _structs(C.c_double)


def build(force=False):
    so = os.path.join(ORACLE_DIR, "libgsr_oracle.so")
    others = [os.path.join(ORACLE_DIR, f) for f in ("libgsr_oracle_fma.so", "libgsr_oracle_f64.so")]
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if force or not all(os.path.exists(q) for q in [so] + others) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


_FMA = False


class fma_twin:
    """Context manager: inside it, oracle.Forward & co. run the FMA-contracted build of the same sources (libgsr_oracle_fma.so).
    |twin - oracle| on a case is the noise floor of the reference's own formulas in fp32 (nvcc contracts by default)."""

    def __enter__(self):
        global _LIB, _FMA
        self.saved = (_LIB, _FMA)
        _LIB, _FMA = None, True
        return self

    def __exit__(self, *a):
        global _LIB, _FMA
        _LIB, _FMA = self.saved


_LIB64 = None


def lib64():
    """libgsr_oracle_f64.so: the same sources with the rasterizer's arithmetic in double (the truth of the parity tests)."""
    global _LIB64
    if _LIB64 is None:
        build()
        L = C.CDLL(os.path.join(ORACLE_DIR, "libgsr_oracle_f64.so"))
        dp = C.POINTER(C.c_double)
        L.ref_forward.restype = C.c_void_p
        L.ref_forward.argtypes = [C.c_int, C.POINTER(RefInputs64), dp, _i32p, dp, _i32p, dp, dp]
        L.ref_backward.restype = None
        L.ref_backward.argtypes = [C.c_void_p, C.POINTER(RefInputs64), C.POINTER(RefOutGrads64), C.POINTER(RefInGrads64)]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_num_rendered.argtypes = [C.c_void_p]; L.ref_num_rendered.restype = C.c_int32
        L.ref_num_tiles.argtypes = [C.c_void_p]; L.ref_num_tiles.restype = C.c_int32
        L.ref_get_image_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_get_geom.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        _LIB64 = L
    return _LIB64


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        if _FMA:
            so = os.path.join(ORACLE_DIR, "libgsr_oracle_fma.so")
        L = C.CDLL(so)
        L.ref_forward.restype = C.c_void_p
        L.ref_forward.argtypes = [C.c_int, C.POINTER(RefInputs), _fp, C.POINTER(C.c_int32), _fp,
                                  C.POINTER(C.c_int32), _fp, _fp]
        L.ref_backward.restype = None
        L.ref_backward.argtypes = [C.c_void_p, C.POINTER(RefInputs), C.POINTER(RefOutGrads), C.POINTER(RefInGrads)]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_num_rendered.argtypes = [C.c_void_p]; L.ref_num_rendered.restype = C.c_int32
        L.ref_num_tiles.argtypes = [C.c_void_p]; L.ref_num_tiles.restype = C.c_int32
        for n in ("ref_get_point_list", "ref_get_keys", "ref_get_ranges", "ref_get_tiles_touched"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        L.ref_get_geom.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.ref_get_image_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_visible_filter.argtypes = [C.POINTER(RefInputs), C.POINTER(C.c_int32)]
        L.ref_mark_visible.argtypes = [C.c_int32, _fp, _fp, _fp, C.c_void_p]
        L.ref_tsdf_integrate.argtypes = [C.c_int64, _fp, _fp, C.c_int32, C.c_int32, _fp, _fp, C.c_float, _fp,
                                         _fp, _fp, _fp]
        L.ref_tsdf_integrate_dense.argtypes = [C.c_int32] * 3 + [_fp, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _fp, _fp,
                                               C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _fp, _fp]
        L.ref_dist2.argtypes = [C.c_int32, _fp, _fp]
        L.ref_omp_threads.restype = C.c_int32
        _LIB = L
    return _LIB


def _f32(a):
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    """float32 VALUES widened to float64: the truth run starts from exactly the numbers the float32 implementations get."""
    a = _f32(a)
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_double) if a.dtype == np.float64 else _fp)


class _Keep:
    """Holds numpy arrays alive next to the ctypes struct that points into them."""

    def __init__(self, wide=False):
        self.arrs = []
        self.wide = wide

    def f(self, a):
        a = _f64(a) if self.wide else _f32(a)
        if a is not None:
            self.arrs.append(a)
        return _p(a)


def make_inputs(scene, variant, wide=False):
    """scene: dict with the keyword tensors of the reference rasterizer call (see tests/scenes.py).  wide: the float64 build's struct."""
    k = _Keep(wide)
    ri = RefInputs64() if wide else RefInputs()
    means3D = _f32(scene["means3D"])
    ri.P = means3D.shape[0]
    shs = scene.get("shs")
    ri.M = 0 if shs is None else int(np.asarray(shs).shape[1])
    ri.D = int(scene.get("sh_degree", 0))
    ri.W = int(scene["W"]); ri.H = int(scene["H"])
    ri.tanfovx = float(np.float32(scene["tanfovx"])); ri.tanfovy = float(np.float32(scene["tanfovy"]))     # the float32 values in both builds
    ri.scale_modifier = float(np.float32(scene.get("scale_modifier", 1.0)))
    ri.prefiltered = 0
    ri.render_geo = int(bool(scene.get("render_geo", True)))
    ri.bg = k.f(scene["bg"]); ri.viewmatrix = k.f(scene["viewmatrix"]); ri.projmatrix = k.f(scene["projmatrix"])
    ri.campos = k.f(scene["campos"]); ri.means3D = k.f(means3D)
    ri.shs = k.f(shs); ri.colors_precomp = k.f(scene.get("colors_precomp"))
    ri.opacities = k.f(scene["opacities"]); ri.scales = k.f(scene.get("scales"))
    ri.rotations = k.f(scene.get("rotations")); ri.cov3D_precomp = k.f(scene.get("cov3D_precomp"))
    ri.all_map = k.f(scene.get("all_map")) if variant == PLANE else None
    ri._keep = k
    return ri


class Forward:
    """Runs the oracle forward; keeps the state for backward/introspection.  Use as a context manager or call .close()."""

    def __init__(self, scene, variant):
        if isinstance(variant, str):
            variant = VARIANT_ID[variant]
        L = lib()
        self.variant = variant
        self.ri = make_inputs(scene, variant)
        P, W, H = self.ri.P, self.ri.W, self.ri.H
        self.P, self.W, self.H = P, W, H
        self.color = np.zeros((3, H, W), np.float32)
        self.radii = np.zeros((P,), np.int32)
        self.others = np.zeros((11, H, W), np.float32) if variant == SURFEL else None
        self.observe = np.zeros((P,), np.int32) if variant == PLANE else None
        self.out_all_map = np.zeros((5, H, W), np.float32) if variant == PLANE else None
        self.plane_depth = np.zeros((1, H, W), np.float32) if variant == PLANE else None
        self.st = L.ref_forward(variant, C.byref(self.ri), _p(self.color), self.radii.ctypes.data_as(C.POINTER(C.c_int32)),
                                _p(self.others),
                                self.observe.ctypes.data_as(C.POINTER(C.c_int32)) if self.observe is not None else None,
                                _p(self.out_all_map), _p(self.plane_depth))
        self.R = L.ref_num_rendered(self.st)
        self.T = L.ref_num_tiles(self.st)

    def point_list(self):
        out = np.zeros((self.R,), np.uint32); lib().ref_get_point_list(self.st, out.ctypes.data); return out

    def pair_counts(self):
        """-> (evaluated, contributing) (pixel, splat) pairs of the forward."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        f = lib().ref_get_pair_counts; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]; f.restype = None
        f(self.st, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def instance_max_alpha(self):
        """[R] float32: per instance of the sorted list, the largest alpha any in-image pixel of its tile sees under the reference's per-pixel gates."""
        out = np.zeros((max(self.R, 1),), np.float32)
        fn = lib().ref_instance_max_alpha; fn.argtypes = [C.c_void_p, C.POINTER(RefInputs), C.c_void_p]; fn.restype = None
        fn(self.st, C.byref(self.ri), out.ctypes.data)
        return out[:self.R]

    def keys(self):
        out = np.zeros((self.R,), np.uint64); lib().ref_get_keys(self.st, out.ctypes.data); return out

    def ranges(self):
        out = np.zeros((self.T, 2), np.uint32); lib().ref_get_ranges(self.st, out.ctypes.data); return out

    def tiles_touched(self):
        out = np.zeros((self.P,), np.uint32); lib().ref_get_tiles_touched(self.st, out.ctypes.data); return out

    def geom(self):
        P = self.P
        d = np.zeros((P,), np.float32); m = np.zeros((P, 2), np.float32); co = np.zeros((P, 4), np.float32)
        rgb = np.zeros((P, 3), np.float32); cov = np.zeros((P, 9 if self.variant == SURFEL else 6), np.float32)
        lib().ref_get_geom(self.st, d.ctypes.data, m.ctypes.data, co.ctypes.data, rgb.ctypes.data, cov.ctypes.data)
        return dict(depths=d, means2D=m, conic_opacity=co, rgb=rgb, cov=cov)

    def image_state(self):
        N = self.W * self.H
        k = 3 if self.variant == SURFEL else 1
        k2 = 2 if self.variant == SURFEL else 1
        ft = np.zeros((k, self.H, self.W), np.float32); nc = np.zeros((k2, self.H, self.W), np.uint32)
        lib().ref_get_image_state(self.st, ft.ctypes.data, nc.ctypes.data)
        return ft, nc

    def backward(self, dL_dcolor=None, dL_dothers=None, dL_dout_all_map=None, dL_dplane_depth=None):
        P, M = self.P, self.ri.M
        surf = self.variant == SURFEL
        k = _Keep()
        og = RefOutGrads(k.f(dL_dcolor), k.f(dL_dothers), k.f(dL_dout_all_map), k.f(dL_dplane_depth))
        g = dict(
            dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dmeans2D=np.zeros((P, 3), np.float32),
            dL_dmeans2D_abs=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
            dL_dopacity=np.zeros((P, 1), np.float32), dL_dcov3D=np.zeros((P, 9 if surf else 6), np.float32),
            dL_dsh=np.zeros((P, max(M, 1), 3), np.float32), dL_dscales=np.zeros((P, 2 if surf else 3), np.float32),
            dL_drotations=np.zeros((P, 4), np.float32), dL_dall_map=np.zeros((P, 5), np.float32),
            dL_dconic=np.zeros((P, 3 if surf else 4), np.float32))
        ig = RefInGrads(*[_p(g[n]) for n, _ in RefInGrads._fields_])
        lib().ref_backward(self.st, C.byref(self.ri), C.byref(og), C.byref(ig))
        if M == 0:
            g["dL_dsh"] = np.zeros((P, 0, 3), np.float32)
        return g

    def close(self):
        if self.st:
            lib().ref_free(self.st); self.st = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Truth:
    """The float64 build of the same oracle sources on the same inputs, walking the INTEGER stages of a float32 run `f32` (an oracle.Forward:
    culling decisions, radii, sorted instance list, tile ranges), so that every image / gradient element is the float64 value of exactly the
    computation the float32 implementations perform.  Besides the outputs it reports how robust the discrete gate decisions of every pixel
    are against float32 rounding (gsr_oracle.h ref_inputs.gate_margin):
        margin  [H,W]  min over the pixel's gate decisions of |value - threshold| / (first-order float32 error bound); > 1 = robust
        gate    [H,W]  which decision is the closest one (GATE_NAMES), splat [H,W] the Gaussian it concerns
        splat_noise [P] max over a Gaussian's evaluated pairs of the relative float32 error bound of its alpha (conditioning of the splat)"""

    def __init__(self, scene, variant, f32, f32_geometry=False):
        """f32_geometry: blend (forward and backward, in float64) the per-gaussian state of the FLOAT32 run `f32` -- transMat / conic, normal, opacity,
        projected centre, depth, SH colour as a float32 preprocess leaves them in the geomBuffer -- instead of this run's own float64 values: the
        distance of such a run from the plain truth is the error floor the reference's float32 preprocess imposes on ANY blend implementation."""
        if isinstance(variant, str):
            variant = VARIANT_ID[variant]
        L = lib64()
        self.variant = variant
        self.ri = ri = make_inputs(scene, variant, wide=True)
        P, W, H = ri.P, ri.W, ri.H
        self.P, self.W, self.H = P, W, H
        self._ov = (np.ascontiguousarray(f32.radii, np.int32), np.ascontiguousarray(f32.point_list(), np.uint32), np.ascontiguousarray(f32.ranges(), np.uint32))
        ri.ov_radii = self._ov[0].ctypes.data_as(_i32p); ri.ov_point_list = self._ov[1].ctypes.data_as(_u32p)
        ri.ov_ranges = self._ov[2].ctypes.data_as(_u32p); ri.ov_R = int(self._ov[1].shape[0])
        if f32_geometry:
            gm = f32.geom()
            self._ovg = {k: np.ascontiguousarray(gm[k], np.float64) for k in ("cov", "conic_opacity", "means2D", "depths", "rgb")}
            ri.ov_cov = _p(self._ovg["cov"]); ri.ov_conic_opacity = _p(self._ovg["conic_opacity"]); ri.ov_means2D = _p(self._ovg["means2D"])
            ri.ov_depths = _p(self._ovg["depths"]); ri.ov_rgb = _p(self._ovg["rgb"])
        self.margin = np.zeros((H, W), np.float64); self.gate = np.zeros((H, W), np.int32); self.splat = np.zeros((H, W), np.int32)
        self.splat_noise = np.zeros((P,), np.float64)
        ri.gate_margin = _p(self.margin); ri.gate_id = self.gate.ctypes.data_as(_i32p); ri.gate_splat = self.splat.ctypes.data_as(_i32p)
        ri.splat_noise = _p(self.splat_noise)
        self.color = np.zeros((3, H, W), np.float64)
        self.radii = np.zeros((P,), np.int32)
        self.others = np.zeros((11, H, W), np.float64) if variant == SURFEL else None
        self.observe = np.zeros((P,), np.int32) if variant == PLANE else None
        self.out_all_map = np.zeros((5, H, W), np.float64) if variant == PLANE else None
        self.plane_depth = np.zeros((1, H, W), np.float64) if variant == PLANE else None
        self.st = L.ref_forward(variant, C.byref(ri), _p(self.color), self.radii.ctypes.data_as(_i32p), _p(self.others),
                                self.observe.ctypes.data_as(_i32p) if self.observe is not None else None, _p(self.out_all_map), _p(self.plane_depth))

    def image_state(self):
        k = 3 if self.variant == SURFEL else 1
        k2 = 2 if self.variant == SURFEL else 1
        ft = np.zeros((k, self.H, self.W), np.float64); nc = np.zeros((k2, self.H, self.W), np.uint32)
        lib64().ref_get_image_state(self.st, ft.ctypes.data, nc.ctypes.data)
        return ft, nc

    def backward(self, dL_dcolor=None, dL_dothers=None, dL_dout_all_map=None, dL_dplane_depth=None):
        P, M = self.P, self.ri.M
        surf = self.variant == SURFEL
        k = _Keep(True)
        og = RefOutGrads64(k.f(dL_dcolor), k.f(dL_dothers), k.f(dL_dout_all_map), k.f(dL_dplane_depth))
        z = lambda *shape: np.zeros(shape, np.float64)
        g = dict(dL_dmeans3D=z(P, 3), dL_dmeans2D=z(P, 3), dL_dmeans2D_abs=z(P, 3), dL_dcolors=z(P, 3), dL_dopacity=z(P, 1),
                 dL_dcov3D=z(P, 9 if surf else 6), dL_dsh=z(P, max(M, 1), 3), dL_dscales=z(P, 2 if surf else 3), dL_drotations=z(P, 4),
                 dL_dall_map=z(P, 5), dL_dconic=z(P, 3 if surf else 4))
        ig = RefInGrads64(*[_p(g[n]) for n, _ in RefInGrads64._fields_])
        lib64().ref_backward(self.st, C.byref(self.ri), C.byref(og), C.byref(ig))
        if M == 0:
            g["dL_dsh"] = z(P, 0, 3)
        return g

    def instance_max_alpha(self):
        """[R] float64: as oracle.Forward.instance_max_alpha, evaluated by the float64 build over the float32 run's list."""
        R = int(self._ov[1].shape[0])
        out = np.zeros((max(R, 1),), np.float64)
        fn = lib64().ref_instance_max_alpha; fn.argtypes = [C.c_void_p, C.POINTER(RefInputs64), C.c_void_p]; fn.restype = None
        fn(self.st, C.byref(self.ri), out.ctypes.data)
        return out[:R]

    def fragile(self):
        """[H,W] bool: pixels where some gate decision is within its float32 error bound of flipping."""
        return self.margin <= 1.0

    def close(self):
        if self.st:
            lib64().ref_free(self.st); self.st = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def visible_filter(scene):
    ri = make_inputs(scene, EWA)
    radii = np.zeros((ri.P,), np.int32)
    lib().ref_visible_filter(C.byref(ri), radii.ctypes.data_as(C.POINTER(C.c_int32)))
    return radii


def mark_visible(means3D, viewmatrix, projmatrix):
    m, v, p = _f32(means3D), _f32(viewmatrix), _f32(projmatrix)
    out = np.zeros((m.shape[0],), np.uint8)
    lib().ref_mark_visible(m.shape[0], _p(m), _p(v), _p(p), out.ctypes.data)
    return out.astype(bool)


def tsdf_integrate(points, full_proj, depth, rgb, sdf_trunc, tsdf, weight, rgb_acc, trunc_pp=None):
    pts, F, d, c = _f32(points), _f32(full_proj), _f32(depth), _f32(rgb)
    H, W = d.shape[-2], d.shape[-1]
    tp = _f32(trunc_pp)
    assert tsdf.dtype == np.float32 and weight.dtype == np.float32 and rgb_acc.dtype == np.float32
    lib().ref_tsdf_integrate(pts.shape[0], _p(pts), _p(F), W, H, _p(d), _p(c), float(sdf_trunc), _p(tp),
                             _p(tsdf), _p(weight), _p(rgb_acc))


def tsdf_integrate_dense(dims, origin, voxel_length, sdf_trunc, depth_trunc, depth, rgb, fx, fy, cx, cy, extrinsic, tsdf, weight, color):
    o, d, c, E = _f32(origin), _f32(depth), _f32(rgb), _f32(extrinsic)
    H, W = d.shape[-2], d.shape[-1]
    lib().ref_tsdf_integrate_dense(int(dims[0]), int(dims[1]), int(dims[2]), _p(o), float(voxel_length), float(sdf_trunc),
                                   float(depth_trunc), W, H, _p(d), _p(c), float(fx), float(fy), float(cx), float(cy), _p(E),
                                   _p(tsdf), _p(weight), _p(color))


def dist2(points):
    p = _f32(points)
    out = np.zeros((p.shape[0],), np.float32)
    lib().ref_dist2(p.shape[0], _p(p), _p(out))
    return out


def omp_threads():
    return int(lib().ref_omp_threads())


def set_threads(n):
    lib().ref_set_threads(C.c_int32(int(n)))


def set_tile_stride(n):
    lib().ref_set_tile_stride(C.c_int32(int(n)))


def loss_l1_ssim(img, gt, lam):
    """-> (loss[3] = {l1, ssim, loss}, dL_dimg) of the photometric loss (oracle/gsl_oracle.c)."""
    L = lib()
    L.ref_loss_l1_ssim.restype = None
    img = _f32(img); gt = _f32(gt)
    Cc, H, W = img.shape
    out = np.zeros(3, np.float32); d = np.zeros_like(img)
    L.ref_loss_l1_ssim(C.c_int32(Cc), C.c_int32(H), C.c_int32(W), _p(img), _p(gt), C.c_float(lam), _p(out), _p(d))
    return out, d


def loss_surfel_geo(allmap, ray_mat, normal_rot, depth_ratio, lambda_normal, lambda_dist):
    """-> dict(loss[3], dL_dallmap, surf_depth, normal_world, surf_normal) (oracle/gsl_oracle.c ref_loss_surfel_geo)."""
    L = lib()
    L.ref_loss_surfel_geo.restype = None
    am = _f32(allmap); rm = _f32(ray_mat).reshape(-1); nr = _f32(normal_rot).reshape(-1)
    _, H, W = am.shape
    out = {"loss": np.zeros(3, np.float32), "dL_dallmap": np.zeros_like(am), "surf_depth": np.zeros((1, H, W), np.float32),
           "normal_world": np.zeros((3, H, W), np.float32), "surf_normal": np.zeros((3, H, W), np.float32)}
    L.ref_loss_surfel_geo(C.c_int32(H), C.c_int32(W), _p(am), _p(rm), _p(nr), C.c_float(depth_ratio), C.c_float(lambda_normal),
                          C.c_float(lambda_dist), _p(out["loss"]), _p(out["dL_dallmap"]), _p(out["surf_depth"]), _p(out["normal_world"]),
                          _p(out["surf_normal"]))
    return out


def loss_plane_geo(depth, alpha, normal, weight, ray_mat, lam):
    """-> dict(loss[3], dL_ddepth, dL_dnormal, depth_normal) (oracle/gsl_oracle.c ref_loss_plane_geo)."""
    L = lib()
    L.ref_loss_plane_geo.restype = None
    d = _f32(depth); a = _f32(alpha); n = _f32(normal); w = None if weight is None else _f32(weight); rm = _f32(ray_mat).reshape(-1)
    H, W = d.shape[-2], d.shape[-1]
    out = {"loss": np.zeros(3, np.float32), "dL_ddepth": np.zeros((H, W), np.float32), "dL_dnormal": np.zeros((3, H, W), np.float32),
           "depth_normal": np.zeros((3, H, W), np.float32)}
    L.ref_loss_plane_geo(C.c_int32(H), C.c_int32(W), _p(d), _p(a), _p(n), _p(w), _p(rm), C.c_float(lam), _p(out["loss"]), _p(out["dL_ddepth"]),
                         _p(out["dL_dnormal"]), _p(out["depth_normal"]))
    return out


class SparseTSDF:
    """oracle/gsr_oracle.c ref_tsdf_sparse_*: CPU restatement of the block-sparse (Open3D ScalableTSDFVolume-style) integration."""

    def __init__(self, voxel_length, sdf_trunc):
        L = lib()
        L.ref_tsdf_sparse_new.restype = C.c_void_p; L.ref_tsdf_sparse_new.argtypes = [C.c_float, C.c_float]
        L.ref_tsdf_sparse_free.argtypes = [C.c_void_p]
        L.ref_tsdf_sparse_integrate.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp,
                                                C.c_float, C.c_int32]
        L.ref_tsdf_sparse_num_units.restype = C.c_int32; L.ref_tsdf_sparse_num_units.argtypes = [C.c_void_p]
        L.ref_tsdf_sparse_get.argtypes = [C.c_void_p] * 5
        self.L = L
        self.h = L.ref_tsdf_sparse_new(float(voxel_length), float(sdf_trunc))

    def integrate(self, rgb, depth, fx, fy, cx, cy, extrinsic, depth_trunc=3.0e38, stride=4):
        d, c = _f32(depth), _f32(rgb)
        H, W = d.shape[-2], d.shape[-1]
        E = np.asarray(extrinsic, np.float64).reshape(4, 4)
        Ea = np.ascontiguousarray(E[:3].reshape(-1), np.float32); Pa = np.ascontiguousarray(np.linalg.inv(E)[:3].reshape(-1), np.float32)
        self.L.ref_tsdf_sparse_integrate(self.h, W, H, _p(d), _p(c), float(fx), float(fy), float(cx), float(cy), _p(Ea), _p(Pa),
                                         float(min(depth_trunc, 3.0e38)), int(stride))

    def units(self):
        n = self.L.ref_tsdf_sparse_num_units(self.h)
        co = np.zeros((n, 3), np.int32); t = np.zeros((n, 16, 16, 16), np.float32); w = np.zeros_like(t); c = np.zeros((n, 16, 16, 16, 3), np.float32)
        if n:
            self.L.ref_tsdf_sparse_get(self.h, co.ctypes.data, t.ctypes.data, w.ctypes.data, c.ctypes.data)
        return co, t, w, c

    def __del__(self):
        try:
            self.L.ref_tsdf_sparse_free(self.h)
        except Exception:
            pass
