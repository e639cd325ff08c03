"""Shared autograd bridge for the three rasterizer variants (the role of _RasterizeGaussians in
diff_gaussian_rasterization/__init__.py:44-155, diff_surfel_rasterization/__init__.py:44-156,
diff_plane_rasterization/__init__.py:48-171) on top of the C ABI of libgsrast_hip.so.

Buffer ownership mirrors the reference: outputs and the three opaque byte arenas (geomBuffer, binningBuffer,
imgBuffer) are torch tensors allocated here; the arenas are saved for backward.
"""
import ctypes as C
import os
import threading

import torch

from . import (EWA, SURFEL, PLANE, Cfg, Inputs, Outputs, OutGrads, InGrads, lib, check, stream_ptr, dev_f32, ptr,
               make_cfg)


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


# Expected number of tile instances per (device, variant, W, H): sizes the binning arena of the NEXT forward so that
# stage 2 can be enqueued without waiting for the host to learn num_rendered (gsr_forward).
# Policy (per key): the hint is a slowly decaying running maximum of the instance counts seen (max(R, 0.9 * previous hint)), so
# alternating cameras / scenes with different counts at one resolution do not overflow on every other call; after two overflows
# in a row the key falls back to the reference-shaped stage1 -> sync -> stage2 path for the next 8 calls.  `capacity_hint` lets a
# caller that knows better pass the expected count itself.  Guarded by a lock: the dicts are the only Python-side shared state.
_R_HINT = {}
_R_OVERFLOWS = {}     # key -> [consecutive overflows, calls left on the exact path]
_HINT_LOCK = threading.Lock()
_SPECULATIVE = True           # (tests flip it to walk the exact stage1 -> sync -> stage2 path; it was the environment switch GSR_SPECULATIVE until round 6)


def _bytes(n, device):
    return torch.empty((max(int(n), 1),), dtype=torch.uint8, device=device)


# ---- forward without any host synchronisation (gsr_forward_async): the form a HIP graph can record.  Selected inside `static_capacity(...)`
# and automatically while the current stream is being captured (torch.cuda.graph).  The binning arena has a FIXED capacity -- the argument, or
# 1.25 x the running-max hint of earlier (eager) calls -- and the host never learns num_rendered: every such call appends (status, capacity)
# to a registry the caller inspects whenever it synchronises anyway: status[0] = instances of the last run, status[1] != 0 = a run
# overflowed the capacity (outputs of that run were incomplete -> raise the capacity and re-run / re-capture; gsrast.graphs does that).
_ASYNC = threading.local()
PREFILTERED_MSG = "Point is filtered although prefiltered is set. This shouldn't happen!"      # auxiliary.h:157; status word 2 of a sync-free forward
_ASYNC_STATUS = []            # [(status int32[3] tensor, capacity)] of the async forwards issued since async_status_reset()


class static_capacity:
    """with static_capacity(cap=None): rasterizer forwards inside run sync-free against a binning arena of `cap` tile instances."""

    def __init__(self, cap=None):
        self.cap = cap

    def __enter__(self):
        self.prev = getattr(_ASYNC, "cap", False)
        _ASYNC.cap = self.cap if self.cap is not None else True
        return self

    def __exit__(self, *a):
        _ASYNC.cap = self.prev


# Status words of forwards recorded INTO A HIP GRAPH live outside the graph: rows of a per-device pool that was zeroed eagerly (the eager run that
# seeds the capacity hint creates it).  The library only ever sets words 1 and 2, so an overflow in ANY replay stays visible until the owner of the
# graph looks (gsrast.graphs.GraphedStep.check) -- a torch.zeros() inside the capture would be a fill node that clears it again on every replay.
_STATUS_POOL = {}             # device index -> [int32 (rows, 3) tensor, rows handed out]
_STATUS_ROWS = 256


def _status_pool(dev):
    """Called on every forward outside a capture: makes sure the device has a pool with free rows."""
    st = _STATUS_POOL.get(dev.index)
    if st is None or st[1] >= _STATUS_ROWS:
        _STATUS_POOL[dev.index] = [torch.zeros((_STATUS_ROWS, 3), dtype=torch.int32, device=dev), 0]


def _status_row(dev):
    st = _STATUS_POOL.get(dev.index)
    if st is None or st[1] >= _STATUS_ROWS:
        raise RuntimeError("gsrast: no zeroed status row for a rasterizer forward under graph capture: run the call once eagerly first "
                           f"(more than {_STATUS_ROWS} forwards recorded since the last eager one)")
    st[1] += 1
    return st[0][st[1] - 1]


def async_status(reset=False):
    """-> list of (num_rendered, overflowed, capacity) for the sync-free forwards recorded so far (reads the device words: synchronises)."""
    host = [(t.cpu(), c) for t, c in _ASYNC_STATUS]
    out = [(int(st[0]), bool(st[1]), cap) for st, cap in host]
    bad = any(bool(st[2]) for st, _ in host)
    if reset:
        _ASYNC_STATUS.clear()
    if bad:
        raise RuntimeError(PREFILTERED_MSG)
    return out


def async_status_reset():
    _ASYNC_STATUS.clear()


# ---- gradient-accumulator scratch of the backward, kept per (device, stream, variant) and left zeroed by the preprocess backward
# (gsr_backward_ex GSR_BWD_SCRATCH_IS_ZERO | GSR_BWD_LEAVE_ZERO): no 24 MB memset + launch gap per iteration.  One buffer serves every P up to
# its size (a call touches -- and clears -- rows [0, P) only; the decode's output row count changes from iteration to iteration), it is replaced
# by a larger one when P outgrows it.
_ACC_REUSE = True
_ACC_CACHE = {}
_ACC_LOCK = threading.Lock()


def _acc_scratch(L, variant, P, dev):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, variant)
    need = max(int(L.gsr_backward_scratch_bytes(variant, P)), 1)
    with _ACC_LOCK:
        t = _ACC_CACHE.pop(key, None)          # taken out while in use: a concurrent backward on the same key gets its own buffer
    if t is None or t.numel() < need:
        t = torch.zeros((need + need // 4,), dtype=torch.uint8, device=dev)      # 25 % head-room: P drifts a little every iteration
    return key, t


def _acc_release(key, t):
    with _ACC_LOCK:
        old = _ACC_CACHE.get(key)
        if old is None or old.numel() <= t.numel():
            _ACC_CACHE[key] = t


def _prepare(variant, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, all_map, settings):
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    keep = []
    m3 = dev_f32(means3D, "means3D", allow_empty=True)
    P = int(means3D.size(0))
    t = dict(means3D=m3, shs=dev_f32(sh, "sh"), colors_precomp=dev_f32(colors_precomp, "colors"),
             opacities=dev_f32(opacities, "opacity"), scales=dev_f32(scales, "scales"),
             rotations=dev_f32(rotations, "rotations"), cov3D_precomp=dev_f32(cov3Ds_precomp, "cov3D_precomp"),
             all_map=dev_f32(all_map, "all_map") if variant == PLANE else None)
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    render_geo = bool(getattr(settings, "render_geo", False)) and t["all_map"] is not None
    cfg = make_cfg(variant, P, settings, settings.sh_degree, M, render_geo, keep)
    inp = Inputs(*[ptr(t[n]) for n, _ in Inputs._fields_])
    keep.extend(v for v in t.values() if v is not None)
    return cfg, inp, keep, P, M


def forward(variant, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, all_map, settings,
            capacity_hint=None):
    """Returns (num_rendered, outputs dict, radii, geomBuffer, binningBuffer, imgBuffer)."""
    L = lib()
    dev = means3D.device
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")     # rasterize_points.cu:57-59
    if dev.type != "cuda":
        raise RuntimeError("means3D must be a CUDA tensor")
    H, W = int(settings.image_height), int(settings.image_width)
    cfg, inp, keep, P, M = _prepare(variant, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                    all_map, settings)
    f32 = dict(dtype=torch.float32, device=dev)
    outs = {}
    empty_call = (P == 0)
    mk = torch.zeros if empty_call else torch.empty
    outs["color"] = mk((3, H, W), **f32)
    radii = mk((P,), dtype=torch.int32, device=dev)          # the preprocess kernel writes every entry
    if variant == SURFEL:
        outs["others"] = mk((11, H, W), **f32)
    if variant == PLANE:
        outs["observe"] = torch.zeros((P,), dtype=torch.int32, device=dev)
        geo = bool(cfg.render_geo) and not empty_call
        outs["all_map"] = (torch.empty if geo else torch.zeros)((5, H, W), **f32)
        outs["plane_depth"] = (torch.empty if geo else torch.zeros)((1, H, W), **f32)
    geom = _bytes(L.gsr_geom_bytes(variant, P), dev)
    img = _bytes(L.gsr_img_bytes(variant, W, H), dev)
    if empty_call:
        return 0, outs, radii, geom, _bytes(0, dev), img
    s = stream_ptr(dev)
    R = C.c_uint32(0)
    o = Outputs(ptr(outs["color"]), ptr(outs.get("others")), ptr(outs.get("observe")), ptr(outs.get("all_map")),
                ptr(outs.get("plane_depth")))
    key = (dev.index, variant, W, H)
    hint = None
    if _SPECULATIVE and not cfg.debug:
        with _HINT_LOCK:
            st = _R_OVERFLOWS.get(key)
            if st is not None and st[1] > 0:
                st[1] -= 1                      # this key overflowed repeatedly: exact path for a while
            else:
                hint = capacity_hint if capacity_hint is not None else _R_HINT.get(key)
    overflowed = False
    acap = getattr(_ASYNC, "cap", False)
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:
        _status_pool(dev)
    elif acap is False:
        acap = True
    if acap is not False and not cfg.debug:
        if acap is True:
            with _HINT_LOCK:
                h = capacity_hint if capacity_hint is not None else _R_HINT.get(key)
            if h is None:
                raise RuntimeError("gsrast: a sync-free forward needs a capacity: pass static_capacity(cap) or run the call once eagerly first")
            acap = int(h * 1.25) + 16384
        binning = _bytes(L.gsr_binning_bytes(variant, int(acap), W, H), dev)
        status = _status_row(dev) if capturing else torch.zeros((3,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(L.gsr_forward_async(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(binning), binning.numel(), ptr(img),
                                      img.numel(), ptr(radii), C.byref(o), ptr(status), s), "forward_async")
        cap = int(L.gsr_binning_capacity(variant, binning.numel(), W, H))
        _ASYNC_STATUS.append((status, cap))
        return cap, outs, radii, geom, binning, img
    with torch.cuda.device(dev):
        if hint is not None:
            # steady state: one call, no GPU idle gap at the sync (arena sized from the previous call + 25 % head-room)
            binning = _bytes(L.gsr_binning_bytes(variant, int(hint * 1.25) + 16384, W, H), dev)
            ovf = C.c_int32(0)
            check(L.gsr_forward(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(binning), binning.numel(), ptr(img),
                                img.numel(), ptr(radii), C.byref(o), C.byref(R), C.byref(ovf), s), "forward")
            if ovf.value:
                overflowed = True
                binning = _bytes(L.gsr_binning_bytes(variant, R.value, W, H), dev)
                if "observe" in outs:
                    outs["observe"].zero_()
                check(L.gsr_forward_stage2(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(binning), binning.numel(),
                                           ptr(img), img.numel(), R.value, C.byref(o), s), "forward")
        else:
            # the depth order stage 1 decided travels back with num_rendered and into stage 2, which is then a pure enqueue (ABI 7; the ABI <= 6 stage 2
            # read the record back from the geom arena: a stream drain per two-stage forward)
            order = C.c_uint32(0)
            check(L.gsr_forward_stage1_ex(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(radii), C.byref(R), C.byref(order), s), "forward")
            binning = _bytes(L.gsr_binning_bytes(variant, R.value, W, H), dev)
            check(L.gsr_forward_stage2_ex(C.byref(cfg), C.byref(inp), ptr(geom), geom.numel(), ptr(binning), binning.numel(),
                                          ptr(img), img.numel(), R.value, order.value, C.byref(o), s), "forward")
    with _HINT_LOCK:
        prev = _R_HINT.get(key)
        _R_HINT[key] = int(R.value) if (prev is None or overflowed) else max(int(R.value), int(0.9 * prev))
        st = _R_OVERFLOWS.setdefault(key, [0, 0])
        if overflowed:
            st[0] += 1
            if st[0] >= 2:
                st[0], st[1] = 0, 8
        elif hint is not None:
            st[0] = 0
    return int(R.value), outs, radii, geom, binning, img


def backward(variant, num_rendered, settings, radii, means3D, sh, colors_precomp, opacities, scales, rotations,
             cov3Ds_precomp, all_map, geom, binning, img, grad_color, grad_others=None, grad_all_map=None,
             grad_plane_depth=None, all_map_pixels=None):
    """Returns dict of gradients (tensors shaped like the reference's RasterizeGaussiansBackwardCUDA outputs)."""
    L = lib()
    dev = means3D.device
    cfg, inp, keep, P, M = _prepare(variant, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                    all_map, settings)
    f32 = dict(dtype=torch.float32, device=dev)
    mk = torch.zeros if P == 0 else torch.empty
    g = dict(dL_dmeans3D=mk((P, 3), **f32), dL_dmeans2D=mk((P, 3), **f32),
             dL_dcolors=mk((P, 3), **f32), dL_dopacity=mk((P, 1), **f32),
             dL_dcov3D=mk((P, 9 if variant == SURFEL else 6), **f32),
             dL_dsh=mk((P, M, 3), **f32), dL_dscales=mk((P, 2 if variant == SURFEL else 3), **f32),
             dL_drotations=mk((P, 4), **f32))
    if variant == PLANE:
        g["dL_dmeans2D_abs"] = mk((P, 3), **f32)
        # without render_geo the all_map channels receive no gradient (PLANE backward.cu:563 is guarded)
        g["dL_dall_map"] = (mk if cfg.render_geo else torch.zeros)((P, 5), **f32)
    if P == 0:
        return g
    def gd(t, name):
        return dev_f32(t, name) if t is not None else None
    og_t = [gd(grad_color, "dL_dout_color"), gd(grad_others, "dL_dout_others"), gd(grad_all_map, "dL_dout_all_map"),
            gd(grad_plane_depth, "dL_dout_plane_depth"), gd(all_map_pixels, "all_map_pixels")]
    og = OutGrads(*[ptr(t) for t in og_t])
    ig = InGrads(ptr(g["dL_dmeans3D"]), ptr(g["dL_dmeans2D"]), ptr(g.get("dL_dmeans2D_abs")), ptr(g["dL_dcolors"]),
                 ptr(g["dL_dopacity"]), ptr(g["dL_dcov3D"]), ptr(g["dL_dsh"]) if M > 0 else None, ptr(g["dL_dscales"]),
                 ptr(g["dL_drotations"]), ptr(g.get("dL_dall_map")))
    radii_c = radii.contiguous()
    if _ACC_REUSE:
        akey, scratch = _acc_scratch(L, variant, P, dev)
        with torch.cuda.device(dev):
            check(L.gsr_backward_ex(C.byref(cfg), C.byref(inp), ptr(radii_c), ptr(geom), geom.numel(), ptr(binning),
                                    binning.numel(), ptr(img), img.numel(), int(num_rendered), ptr(scratch), scratch.numel(),
                                    C.byref(og), C.byref(ig), 3, stream_ptr(dev)), "backward")      # 3 = SCRATCH_IS_ZERO | LEAVE_ZERO
        _acc_release(akey, scratch)             # only after a successful enqueue: a failed call may have left it dirty
    else:
        scratch = _bytes(L.gsr_backward_scratch_bytes(variant, P), dev)
        with torch.cuda.device(dev):
            check(L.gsr_backward(C.byref(cfg), C.byref(inp), ptr(radii_c), ptr(geom), geom.numel(), ptr(binning),
                                 binning.numel(), ptr(img), img.numel(), int(num_rendered), ptr(scratch), scratch.numel(),
                                 C.byref(og), C.byref(ig), stream_ptr(dev)), "backward")
    if sh is None or sh.numel() == 0 or M == 0:
        g["dL_dsh"] = torch.zeros((P, M, 3), **f32)
    return g


def debug_read(variant, field, settings, P, M, num_rendered, geom, binning, img, out):
    keep = []
    cfg = make_cfg(variant, P, settings, settings.sh_degree, M, False, keep)
    check(lib().gsr_debug_read(C.byref(cfg), field, ptr(geom), ptr(binning), binning.numel(), ptr(img), int(num_rendered),
                               ptr(out), stream_ptr(out.device)), "debug_read")
    return out
