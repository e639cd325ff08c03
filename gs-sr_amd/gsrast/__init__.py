"""gsrast -- ctypes binding of libgsrast_hip.so (include/gsrast.h) and the shared autograd bridge used by the
drop-in packages diff_gaussian_rasterization / diff_surfel_rasterization / diff_plane_rasterization /
scaffold_filter / simple_knn that sit next to this package.

There is NO CPU or PyTorch fallback: every entry point raises if the HIP library is missing or a tensor is not on
a HIP device (the reference's extensions are CUDA-only in exactly the same way, e.g. CHECK_INPUT in
diff-surfel-rasterization/rasterize_points.cu:27-29).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libgsrast_hip.so")   # override: experiments only

EWA, SURFEL, PLANE = 0, 1, 2
ABI_VERSION = 8

_vp = C.c_void_p


class Cfg(C.Structure):
    _fields_ = [("variant", C.c_int32), ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32),
                ("H", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
                ("prefiltered", C.c_int32), ("debug", C.c_int32), ("render_geo", C.c_int32),
                ("bg", _vp), ("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp)]


class Inputs(C.Structure):
    _fields_ = [(n, _vp) for n in ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                   "cov3D_precomp", "all_map")]


class Outputs(C.Structure):
    _fields_ = [(n, _vp) for n in ("out_color", "out_others", "out_observe", "out_all_map", "out_plane_depth")]


class OutGrads(C.Structure):
    _fields_ = [(n, _vp) for n in ("dL_dcolor", "dL_dothers", "dL_dout_all_map", "dL_dplane_depth", "all_map_pixels")]


class InGrads(C.Structure):
    _fields_ = [(n, _vp) for n in ("dL_dmeans3D", "dL_dmeans2D", "dL_dmeans2D_abs", "dL_dcolors", "dL_dopacity",
                                   "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dall_map")]


class MvCfg(C.Structure):
    """gsr_mv_cfg (include/gsrast.h)."""
    _fields_ = ([(n, C.c_int32) for n in ("W", "H", "Wn", "Hn", "Wg", "Hg")] +
                [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "nfx", "nfy", "ncx", "ncy")] +
                [("v2n", C.c_float * 12), ("n2v", C.c_float * 12), ("ncc_scale", C.c_float), ("noise_th", C.c_float), ("patch", C.c_int32)])


class TsdfSparse(C.Structure):
    """gsr_tsdf_sparse (include/gsrast.h)."""
    MAX_CHUNKS = 24
    _fields_ = [("keys", _vp), ("slot", _vp), ("coord", _vp), ("stamp", _vp), ("list", _vp), ("counters", _vp), ("mask", _vp), ("chunk", _vp * 24),
                ("chunk0_log2", C.c_uint32), ("n_chunks", C.c_uint32), ("cap_hash_log2", C.c_uint32), ("cap_blocks", C.c_uint32),
                ("voxel_length", C.c_float), ("sdf_trunc", C.c_float)]


class LodCfg(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("fork", C.c_float), ("standard_dist", C.c_float), ("resolution_scale", C.c_float),
                ("coarse_index", C.c_int32), ("mode", C.c_int32)]


EXPORTS = ["gsr_geom_bytes", "gsr_img_bytes", "gsr_binning_bytes", "gsr_backward_scratch_bytes",
           "gsr_forward_stage1", "gsr_forward_stage2", "gsr_forward_stage1_ex", "gsr_forward_stage2_ex", "gsr_backward", "gsr_backward_ex", "gsr_forward_async", "gsr_mark_visible", "gsr_visible_filter",
           "gsr_tsdf_integrate", "gsr_tsdf_integrate_dense", "gsr_tsdf_sparse_integrate2", "gsr_tsdf_sparse_status", "gsr_tsdf_sparse_rehash", "gsr_tsdf_sparse_merge", "gsr_tsdf_sparse_merge_volume", "gsr_tsdf_sparse_materialize", "gsr_loss_l1_linear", "gsr_dist2_scratch_bytes", "gsr_dist2", "gsr_debug_read", "gsr_last_error",
           "gsr_abi_version", "gsr_profile_enable", "gsr_profile_read", "gsr_binning_capacity", "gsr_forward",
           "gsr_loss_l1_ssim_scratch_bytes", "gsr_loss_l1_ssim", "gsr_loss_surfel_geo_scratch_bytes", "gsr_loss_surfel_geo", "gsr_loss_plane_geo", "gsr_loss_scaling_prod", "gsr_octree_visible",
           "gsr_loss_plane_mv_scratch_bytes", "gsr_loss_plane_mv_geo", "gsr_loss_plane_mv_ncc", "gsr_loss_plane_mv_values", "gsr_loss_plane_mv_scale",
           "gsr_plane_allmap", "gsr_plane_allmap_backward", "gsr_gauss_activations", "gsr_gauss_activations_backward", "gsr_sample_mask_scratch_bytes", "gsr_sample_mask", "gsr_densify_stats", "gsr_adam_step", "gsr_adam_step_multi", "gsr_adam_step_multi_dev"]
PROF_LABELS = ["preprocess", "depth_order", "binning", "blend_fwd", "bwd_memset", "blend_bwd", "preprocess_bwd", "_"]

_lib = None


def lib():
    """Loads libgsrast_hip.so; fails loudly when it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"gsrast: HIP library not built: {LIB_PATH} (run `make -C gs-sr_amd/csrc` or "
                           f"`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    sz = C.c_size_t
    L.gsr_geom_bytes.restype = sz; L.gsr_geom_bytes.argtypes = [C.c_int32, C.c_int32]
    L.gsr_img_bytes.restype = sz; L.gsr_img_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.gsr_binning_bytes.restype = sz; L.gsr_binning_bytes.argtypes = [C.c_int32, C.c_uint32, C.c_int32, C.c_int32]
    L.gsr_backward_scratch_bytes.restype = sz; L.gsr_backward_scratch_bytes.argtypes = [C.c_int32, C.c_int32]
    L.gsr_forward_stage1.restype = C.c_int
    L.gsr_forward_stage1.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), _vp, sz, _vp, C.POINTER(C.c_uint32), _vp]
    L.gsr_forward_stage1_ex.restype = C.c_int
    L.gsr_forward_stage1_ex.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), _vp, sz, _vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _vp]
    L.gsr_forward_stage2.restype = C.c_int
    L.gsr_forward_stage2.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), _vp, sz, _vp, sz, _vp, sz, C.c_uint32,
                                     C.POINTER(Outputs), _vp]
    L.gsr_forward_stage2_ex.restype = C.c_int
    L.gsr_forward_stage2_ex.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), _vp, sz, _vp, sz, _vp, sz, C.c_uint32, C.c_uint32,
                                        C.POINTER(Outputs), _vp]
    L.gsr_binning_capacity.restype = C.c_uint32; L.gsr_binning_capacity.argtypes = [C.c_int32, sz, C.c_int32, C.c_int32]
    L.gsr_forward.restype = C.c_int
    L.gsr_forward.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), _vp, sz, _vp, sz, _vp, sz, _vp, C.POINTER(Outputs),
                              C.POINTER(C.c_uint32), C.POINTER(C.c_int32), _vp]
    L.gsr_backward.restype = C.c_int
    L.gsr_backward.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), _vp, _vp, sz, _vp, sz, _vp, sz, C.c_uint32, _vp, sz,
                               C.POINTER(OutGrads), C.POINTER(InGrads), _vp]
    L.gsr_backward_ex.restype = C.c_int
    L.gsr_backward_ex.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), _vp, _vp, sz, _vp, sz, _vp, sz, C.c_uint32, _vp, sz,
                                  C.POINTER(OutGrads), C.POINTER(InGrads), C.c_uint32, _vp]
    L.gsr_forward_async.restype = C.c_int
    L.gsr_forward_async.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), _vp, sz, _vp, sz, _vp, sz, _vp, C.POINTER(Outputs), _vp, _vp]
    L.gsr_mark_visible.restype = C.c_int
    L.gsr_mark_visible.argtypes = [C.c_int32, _vp, _vp, _vp, _vp, _vp]
    L.gsr_visible_filter.restype = C.c_int
    L.gsr_visible_filter.argtypes = [C.POINTER(Cfg), _vp, _vp, _vp, _vp, _vp, _vp]
    L.gsr_tsdf_integrate.restype = C.c_int
    L.gsr_tsdf_integrate.argtypes = [C.c_int64, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]
    L.gsr_tsdf_integrate_dense.restype = C.c_int
    L.gsr_tsdf_integrate_dense.argtypes = [C.c_int32] * 3 + [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                           _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), _vp, _vp, _vp, _vp]
    L.gsr_tsdf_sparse_integrate2.restype = C.c_int
    L.gsr_tsdf_sparse_integrate2.argtypes = [C.POINTER(TsdfSparse), C.c_int32, C.c_int32, _vp, _vp, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                             C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int32, C.c_uint32, _vp, _vp, C.c_uint32, _vp]
    L.gsr_tsdf_sparse_status.restype = C.c_int
    L.gsr_tsdf_sparse_status.argtypes = [C.POINTER(TsdfSparse), _vp, _vp]
    L.gsr_tsdf_sparse_rehash.restype = C.c_int
    L.gsr_tsdf_sparse_rehash.argtypes = [C.POINTER(TsdfSparse), C.c_int32, _vp]
    L.gsr_tsdf_sparse_merge.restype = C.c_int
    L.gsr_tsdf_sparse_merge.argtypes = [C.POINTER(TsdfSparse), C.c_int32, _vp, _vp, _vp, _vp, _vp]
    L.gsr_tsdf_sparse_merge_volume.restype = C.c_int
    L.gsr_tsdf_sparse_merge_volume.argtypes = [C.POINTER(TsdfSparse), C.POINTER(TsdfSparse), C.c_int32, _vp]
    L.gsr_tsdf_sparse_materialize.restype = C.c_int
    L.gsr_tsdf_sparse_materialize.argtypes = [C.POINTER(TsdfSparse), C.c_int32, _vp]
    L.gsr_adam_step_multi.restype = C.c_int
    L.gsr_adam_step_multi.argtypes = [C.c_int32, _vp, _vp]
    L.gsr_adam_step_multi_dev.restype = C.c_int
    L.gsr_adam_step_multi_dev.argtypes = [C.c_int32, _vp, _vp, _vp]
    L.gsr_adam_step.restype = C.c_int
    L.gsr_adam_step.argtypes = [C.c_int64, _vp, _vp, _vp, _vp, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, _vp, _vp]
    L.gsr_loss_l1_linear.restype = C.c_int
    L.gsr_loss_l1_linear.argtypes = [C.c_int64, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp]
    L.gsr_loss_l1_ssim_scratch_bytes.restype = sz; L.gsr_loss_l1_ssim_scratch_bytes.argtypes = [C.c_int32] * 3
    L.gsr_loss_l1_ssim.restype = C.c_int
    L.gsr_loss_l1_ssim.argtypes = [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, C.c_float, _vp, _vp, _vp, sz, _vp]
    L.gsr_loss_surfel_geo_scratch_bytes.restype = sz; L.gsr_loss_surfel_geo_scratch_bytes.argtypes = [C.c_int32] * 2
    L.gsr_loss_surfel_geo.restype = C.c_int
    L.gsr_loss_surfel_geo.argtypes = [C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, sz, _vp]
    L.gsr_loss_scaling_prod.restype = C.c_int
    L.gsr_loss_scaling_prod.argtypes = [C.c_int64, C.c_int32, C.c_int32, _vp, _vp, C.c_float, _vp, _vp, _vp]
    L.gsr_loss_plane_geo.restype = C.c_int
    L.gsr_loss_plane_geo.argtypes = [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, sz, _vp]
    L.gsr_loss_plane_mv_scratch_bytes.restype = sz; L.gsr_loss_plane_mv_scratch_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.gsr_loss_plane_mv_geo.restype = C.c_int
    L.gsr_loss_plane_mv_geo.argtypes = [C.POINTER(MvCfg)] + [_vp] * 9 + [sz, _vp]
    L.gsr_loss_plane_mv_ncc.restype = C.c_int
    L.gsr_loss_plane_mv_ncc.argtypes = [C.POINTER(MvCfg), C.c_int32] + [_vp] * 12 + [sz, _vp]
    L.gsr_loss_plane_mv_values.restype = C.c_int
    L.gsr_loss_plane_mv_values.argtypes = [_vp, C.c_float, C.c_float, _vp, _vp]
    L.gsr_loss_plane_mv_scale.restype = C.c_int
    L.gsr_loss_plane_mv_scale.argtypes = [sz, sz, sz, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]
    L.gsr_sample_mask_scratch_bytes.restype = sz; L.gsr_sample_mask_scratch_bytes.argtypes = [C.c_int64]
    L.gsr_sample_mask.restype = C.c_int
    L.gsr_sample_mask.argtypes = [C.c_int64, _vp, C.c_int32, C.c_uint64, _vp, _vp, sz, _vp]
    L.gsr_densify_stats.restype = C.c_int
    L.gsr_densify_stats.argtypes = [C.c_int32, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.gsr_plane_allmap.restype = C.c_int
    L.gsr_plane_allmap.argtypes = [C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]
    L.gsr_plane_allmap_backward.restype = C.c_int
    L.gsr_plane_allmap_backward.argtypes = [C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]
    L.gsr_gauss_activations.restype = C.c_int
    L.gsr_gauss_activations.argtypes = [C.c_int32, C.c_int32] + [_vp] * 7
    L.gsr_gauss_activations_backward.restype = C.c_int
    L.gsr_gauss_activations_backward.argtypes = [C.c_int32, C.c_int32] + [_vp] * 11
    L.gsr_octree_visible.restype = C.c_int
    L.gsr_octree_visible.argtypes = [C.POINTER(Cfg), C.POINTER(LodCfg), _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]
    L.gsr_dist2_scratch_bytes.restype = sz; L.gsr_dist2_scratch_bytes.argtypes = [C.c_int32]
    L.gsr_dist2.restype = C.c_int
    L.gsr_dist2.argtypes = [C.c_int32, _vp, _vp, _vp, sz, _vp]
    L.gsr_debug_read.restype = C.c_int
    L.gsr_debug_read.argtypes = [C.POINTER(Cfg), C.c_int32, _vp, _vp, sz, _vp, C.c_uint32, _vp, _vp]
    L.gsr_profile_enable.restype = C.c_int; L.gsr_profile_enable.argtypes = [C.c_int32]
    L.gsr_profile_read.restype = C.c_int
    L.gsr_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.gsr_last_error.restype = C.c_char_p
    L.gsr_abi_version.restype = C.c_int32
    if L.gsr_abi_version() != ABI_VERSION:
        raise RuntimeError("gsrast: ABI version mismatch between python binding and libgsrast_hip.so")
    _lib = L
    return L


def profile_enable(on=True, stages=None):
    """Starts (and resets) / stops the stage profiler.  stages: names out of PROF_LABELS to time only those (each timed stage costs ~10 us
    of stream idle time per launch: two HIP event records)."""
    if not on:
        lib().gsr_profile_enable(0)
    elif stages is None:
        lib().gsr_profile_enable(1)
    else:
        lib().gsr_profile_enable(sum(1 << PROF_LABELS.index(n) for n in stages) << 8)


def profile_read():
    """-> {label: (total_ms, launches)} accumulated since profile_enable(True)."""
    ms = (C.c_double * 8)(); n = (C.c_uint64 * 8)()
    lib().gsr_profile_read(ms, n)
    return {PROF_LABELS[i]: (ms[i], int(n[i])) for i in range(7)}


def last_error():
    return lib().gsr_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"gsrast {what}: {last_error()}")


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dev_f32(t, name, allow_empty=True):
    """The reference does x.contiguous().data<float>(): float32 required, empty tensor == 'not provided'."""
    if t is None or t.numel() == 0:
        if not allow_empty:
            raise RuntimeError(f"{name} must be provided")
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")          # CHECK_INPUT message of the surfel extension
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected scalar type Float but found {t.dtype}")
    return t.contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def make_cfg(variant, P, settings, D, M, render_geo, keep):
    cfg = Cfg()
    cfg.variant = variant; cfg.P = P; cfg.D = int(D); cfg.M = int(M)
    cfg.W = int(settings.image_width); cfg.H = int(settings.image_height)
    cfg.tanfovx = float(settings.tanfovx); cfg.tanfovy = float(settings.tanfovy)
    cfg.scale_modifier = float(settings.scale_modifier)
    cfg.prefiltered = int(bool(settings.prefiltered)); cfg.debug = int(bool(settings.debug))
    cfg.render_geo = int(bool(render_geo))
    for name, attr in (("bg", "bg"), ("viewmatrix", "viewmatrix"), ("projmatrix", "projmatrix"), ("campos", "campos")):
        t = dev_f32(getattr(settings, attr), name, allow_empty=False)
        keep.append(t)
        setattr(cfg, name, t.data_ptr())
    return cfg
