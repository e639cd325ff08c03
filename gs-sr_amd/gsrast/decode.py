"""Fused neural-Gaussian decode (include/gsdecode.h) behind a torch.autograd.Function.

Drop-in for the body of ScaffoldScene.generate_neural_gaussians (gssr/scene/scaffold_scene.py:27-120) and
OctreeScene.generate_neural_gaussians (gssr/scene/octree_scene.py:26-133): same inputs, same returned tuple
`(xyz, color, opacity, scaling, rot, neural_opacity, mask)`, same gradients to anchors, features, offsets, scalings, the three MLP heads
and the appearance embedding row.  HIP only: there is no CPU path.
"""
import ctypes as C

import torch

from . import check, dev_f32, lib, ptr, stream_ptr

_vp = C.c_void_p
PARAM_NAMES = ("W1o", "b1o", "W2o", "b2o", "W1c", "b1c", "W2c", "b2c", "W1k", "b1k", "W2k", "b2k", "app")
EXPORTS = ["gsd_compact_scratch_bytes", "gsd_compact_visible", "gsd_compact_visible_padded", "gsd_forward_scratch_bytes", "gsd_forward_stage1", "gsd_forward_stage2",
           "gsd_forward", "gsd_forward_static", "gsd_forward_deferred", "gsd_backward_scratch_bytes", "gsd_backward", "gsd_training_stats_scratch_bytes", "gsd_training_stats"]


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("Na", "Nv", "k", "A", "dist_o", "dist_c", "dist_k", "level")]


class Inputs(C.Structure):
    _fields_ = [(n, _vp) for n in ("anchor", "feat", "offset", "scaling", "level", "opacity_scale", "vis_idx", "campos")]


class Params(C.Structure):
    _fields_ = [(n, _vp) for n in PARAM_NAMES]


class Outputs(C.Structure):
    _fields_ = [(n, _vp) for n in ("xyz", "color", "opacity", "scaling", "rot")]


class OutGrads(C.Structure):
    _fields_ = [(n, _vp) for n in ("xyz", "color", "opacity", "scaling", "rot")]


class InGrads(C.Structure):
    _fields_ = [("anchor", _vp), ("feat", _vp), ("offset", _vp), ("scaling", _vp), ("params", Params)]


_bound = False


def _lib():
    global _bound
    L = lib()
    if not _bound:
        sz = C.c_size_t
        L.gsd_compact_scratch_bytes.restype = sz; L.gsd_compact_scratch_bytes.argtypes = [C.c_int32]
        L.gsd_compact_visible.restype = C.c_int
        L.gsd_compact_visible.argtypes = [_vp, C.c_int32, _vp, C.POINTER(C.c_uint32), _vp, sz, _vp]
        L.gsd_compact_visible_padded.restype = C.c_int
        L.gsd_compact_visible_padded.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, sz, _vp]
        L.gsd_forward_scratch_bytes.restype = sz; L.gsd_forward_scratch_bytes.argtypes = [C.c_int32]
        L.gsd_forward_stage1.restype = C.c_int
        L.gsd_forward_stage1.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), C.POINTER(Params), _vp, _vp, _vp, C.POINTER(C.c_uint32), _vp, sz, _vp]
        L.gsd_forward_stage2.restype = C.c_int
        L.gsd_forward_stage2.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), C.POINTER(Params), _vp, _vp, C.c_uint32, C.POINTER(Outputs), _vp, sz, _vp]
        L.gsd_forward.restype = C.c_int
        L.gsd_forward.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), C.POINTER(Params), _vp, _vp, _vp, C.POINTER(Outputs),
                                  C.POINTER(C.c_uint32), _vp, sz, _vp]
        L.gsd_forward_static.restype = C.c_int
        L.gsd_forward_static.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), C.POINTER(Params), _vp, _vp, _vp, C.POINTER(Outputs), _vp, _vp, sz, _vp]
        L.gsd_forward_deferred.restype = C.c_int
        L.gsd_forward_deferred.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), C.POINTER(Params), _vp, _vp, _vp, C.POINTER(Outputs), _vp, _vp, _vp, sz, _vp]
        L.gsd_backward_scratch_bytes.restype = sz; L.gsd_backward_scratch_bytes.argtypes = [C.POINTER(Cfg)]
        L.gsd_backward.restype = C.c_int
        L.gsd_backward.argtypes = [C.POINTER(Cfg), C.POINTER(Inputs), C.POINTER(Params), _vp, _vp, C.c_uint32, C.POINTER(OutGrads),
                                   C.POINTER(InGrads), _vp, _vp, sz, _vp]
        L.gsd_training_stats_scratch_bytes.restype = sz; L.gsd_training_stats_scratch_bytes.argtypes = [C.c_int32]
        L.gsd_training_stats.restype = C.c_int
        L.gsd_training_stats.argtypes = [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, sz, _vp]
        _bound = True
    return L


def compact_visible(visible_mask, padded=False):
    """nonzero(visible_mask) as int32 indices (ascending), computed on the device.
    padded=False: exact length, one host synchronisation for the count (the reference's boolean indexing synchronises the same way).
    padded=True : NO synchronisation -- the result has Na entries, the visible indices first and -1 behind them; the decode and the
                  statistics kernels skip the padding rows (include/gsdecode.h gsd_compact_visible_padded)."""
    if not visible_mask.is_cuda:
        raise RuntimeError("visible_mask must be a CUDA tensor")
    m = visible_mask.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else (m != 0).view(torch.uint8)
    Na = m.numel()
    L = _lib()
    idx = torch.empty(Na, dtype=torch.int32, device=m.device)
    scratch = torch.empty(L.gsd_compact_scratch_bytes(Na), dtype=torch.uint8, device=m.device)
    if padded:
        check(L.gsd_compact_visible_padded(ptr(m), Na, ptr(idx), None, ptr(scratch), scratch.numel(), stream_ptr(m.device)), "compact_visible")
        return idx
    n = C.c_uint32(0)
    check(L.gsd_compact_visible(ptr(m), Na, ptr(idx), C.byref(n), ptr(scratch), scratch.numel(), stream_ptr(m.device)), "compact_visible")
    return idx[: n.value]


def _structs(flags, Na, Nv, tensors, params):
    k, A, dist_o, dist_c, dist_k, has_level = flags
    cfg = Cfg(Na, Nv, k, A, int(dist_o), int(dist_c), int(dist_k), int(has_level))
    inp = Inputs(*[ptr(tensors[n]) for n in ("anchor", "feat", "offset", "scaling", "level", "opacity_scale", "vis_idx", "campos")])
    prm = Params(*[ptr(params[n]) for n in PARAM_NAMES])
    return cfg, inp, prm


def _decode_inputs(vis_idx, campos, level, opacity_scale, anchor, feat, offset, scaling, params):
    t = {"anchor": dev_f32(anchor, "anchor", False), "feat": dev_f32(feat, "feat", False), "offset": dev_f32(offset, "offset", False),
         "scaling": dev_f32(scaling, "scaling", False), "level": dev_f32(level, "level"), "opacity_scale": dev_f32(opacity_scale, "opacity_scale"),
         "vis_idx": vis_idx.contiguous(), "campos": dev_f32(campos, "campos", False)}
    if t["vis_idx"].dtype != torch.int32:
        raise RuntimeError("vis_idx must be int32")
    prm = {n: dev_f32(p, n) for n, p in zip(PARAM_NAMES, params)}
    return t, prm


def _decode_launch(flags, t, prm, mode):
    """Enqueue the decode forward.  mode "sync": gsd_forward (one host synchronisation, n = P on return); "static": gsd_forward_static (no
    synchronisation, all Nv*k rows, device count); "deferred": gsd_forward_deferred -- the count is copied to pinned memory behind stage 1 with an
    event behind the copy, stage 2 follows; the caller learns P later (PendingDecode.finish).  -> dict of the buffers the autograd node keeps."""
    L = _lib()
    dev = t["anchor"].device
    k = flags[0]
    Na, Nv = t["anchor"].shape[0], t["vis_idx"].numel()
    cfg, inp, cp = _structs(flags, Na, Nv, t, prm)
    b = {"nop": torch.empty(Nv * k, 1, dtype=torch.float32, device=dev), "mask": torch.empty(Nv * k, dtype=torch.uint8, device=dev),
         "row_offset": torch.empty(max(Nv, 1), dtype=torch.int32, device=dev),
         "scratch": torch.empty(L.gsd_forward_scratch_bytes(Nv), dtype=torch.uint8, device=dev), "count": None}
    cap = Nv * k              # worst case: every offset emitted -> stage 2 is enqueued without waiting for the host to learn P
    b["out"] = [torch.empty(cap, c, dtype=torch.float32, device=dev) for c in (3, 3, 1, 3, 4)]       # xyz, color, opacity, scaling, rot
    out = Outputs(*[ptr(x) for x in b["out"]])
    if mode == "sync":
        P = C.c_uint32(0)
        check(L.gsd_forward(C.byref(cfg), C.byref(inp), C.byref(cp), ptr(b["nop"]), ptr(b["mask"]), ptr(b["row_offset"]), C.byref(out), C.byref(P),
                            ptr(b["scratch"]), b["scratch"].numel(), stream_ptr(dev)), "decode forward")
        b["n"] = P.value
        return b
    if mode == "deferred":    # the count leaves for pinned memory right behind stage 1; the event is recorded there, stage 2 follows (gsd_forward_deferred)
        b["slot"] = _count_slot(dev)
        b["count_host"], b["event"] = b["slot"]
        check(L.gsd_forward_deferred(C.byref(cfg), C.byref(inp), C.byref(cp), ptr(b["nop"]), ptr(b["mask"]), ptr(b["row_offset"]), C.byref(out),
                                     b["count_host"].data_ptr(), b["event"].cuda_event, ptr(b["scratch"]), b["scratch"].numel(), stream_ptr(dev)),
              "decode forward (deferred count)")
        b["n"] = cap
        b["t"], b["prm"] = t, prm
        return b
    b["count"] = torch.empty(1, dtype=torch.int32, device=dev)
    check(L.gsd_forward_static(C.byref(cfg), C.byref(inp), C.byref(cp), ptr(b["nop"]), ptr(b["mask"]), ptr(b["row_offset"]), C.byref(out), ptr(b["count"]),
                               ptr(b["scratch"]), b["scratch"].numel(), stream_ptr(dev)), "decode forward (static rows)")
    b["n"] = cap
    return b


_count_slots = {}       # device index -> free (pinned int32[1], event) pairs: a pair is taken per deferred decode and returned by finish()


def _count_slot(dev):
    free = _count_slots.setdefault(dev.index if dev.index is not None else torch.cuda.current_device(), [])
    if free:
        return free.pop()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))          # creates the hipEvent_t (torch makes it lazily); the library re-records it behind the count copy
    return torch.zeros(1, dtype=torch.int32, pin_memory=True), ev


SCAN_POISON = 0xFFFFFFFF      # csrc/gsd_decode.hip GSD_SCAN_POISON: the scan's look-back gave up


def _return_slot(b):
    slot = b.pop("slot", None)
    if slot is not None:
        dev = b["nop"].device
        _count_slots[dev.index if dev.index is not None else torch.cuda.current_device()].append(slot)


class PendingDecode:
    """A decode whose kernels are enqueued but whose Gaussian count the host has not read yet (neural_gaussians(..., deferred=True)).  Enqueue other
    device work -- e.g. the next camera's LOD mask and decode -- then call finish(): it waits for THIS decode's count only (an event behind its
    kernels, not the stream) and returns the reference-shaped tuple (xyz, color, opacity, scaling, rot, neural_opacity, mask) with its autograd node.
    The node saves for its backward exactly the buffers the enqueued kernels read (`launched["t"]`, `launched["prm"]`: the float32 / contiguous copies
    _decode_inputs made at launch time, or the caller's own tensors where no copy was needed) -- not a second conversion of the arguments at finish().
    The caller's tensors must still not be modified in place in between when they were used without a copy.  A PendingDecode that is dropped unfinished
    gives its pinned count slot and event back when it is collected."""

    def __init__(self, flags, args, launched):
        self._flags, self._args, self._launched = flags, args, launched

    def finish(self):
        b = self._launched
        if b is None:
            raise RuntimeError("PendingDecode.finish() called twice")
        self._launched = None
        b["event"].synchronize()
        n = int(b["count_host"][0]) & 0xFFFFFFFF
        _return_slot(b)
        if n == SCAN_POISON:
            raise RuntimeError("decode forward (deferred count): the scan's look-back timed out (a workgroup in front never published its total)")
        b["n"] = n
        return _NeuralDecode.apply(tuple(self._flags[:6]) + (False, b), *self._args)

    def __del__(self):
        b = getattr(self, "_launched", None)
        if b is not None:
            try:
                b["event"].synchronize()      # the copy into the pinned word must have landed before the slot is handed to the next decode
                _return_slot(b)
            except Exception:                 # interpreter shutdown
                pass


class _NeuralDecode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flags, vis_idx, campos, level, opacity_scale, anchor, feat, offset, scaling, *params):
        static_rows = len(flags) > 6 and bool(flags[6])
        launched = flags[7] if len(flags) > 7 else None           # PendingDecode.finish(): the kernels ran already
        flags = tuple(flags[:6])
        if launched is not None:      # the buffers the enqueued kernels read ARE the ones saved for the backward (a second _decode_inputs here made
            t, prm, b = launched["t"], launched["prm"], launched      # a second set of copies for non-contiguous / non-float32 arguments)
        else:
            t, prm = _decode_inputs(vis_idx, campos, level, opacity_scale, anchor, feat, offset, scaling, params)
            b = _decode_launch(flags, t, prm, "static" if static_rows else "sync")
        nop, mask, row_offset, scratch, count, n = b["nop"], b["mask"], b["row_offset"], b["scratch"], b["count"], b["n"]
        xyz, color, opacity, scl, rot = b["out"]
        if not static_rows:
            xyz, color, opacity, scl, rot = xyz[:n], color[:n], opacity[:n], scl[:n], rot[:n]
        ctx.flags, ctx.n = flags, n
        ctx.save_for_backward(t["anchor"], t["feat"], t["offset"], t["scaling"], t["level"], t["opacity_scale"], t["vis_idx"], t["campos"],
                              nop, row_offset, scratch, *[prm[nm] for nm in PARAM_NAMES])
        mask_b = mask.view(torch.bool)
        ctx.set_materialize_grads(False)     # no zero-filled (Nv*k) gradient tensors for nop / mask (or an unused output) on every backward
        if static_rows:
            ctx.mark_non_differentiable(nop, mask_b, count)
            return xyz, color, opacity, scl, rot, nop, mask_b, count
        ctx.mark_non_differentiable(nop, mask_b)
        return xyz, color, opacity, scl, rot, nop, mask_b

    @staticmethod
    def backward(ctx, g_xyz, g_color, g_opacity, g_scaling, g_rot, _g_nop, _g_mask, _g_count=None):
        L = _lib()
        sv = ctx.saved_tensors
        names = ("anchor", "feat", "offset", "scaling", "level", "opacity_scale", "vis_idx", "campos")
        t = dict(zip(names, sv[:8]))
        nop, row_offset, fwd_scratch = sv[8], sv[9], sv[10]
        prm = dict(zip(PARAM_NAMES, sv[11:]))
        dev = t["anchor"].device
        Na, Nv, n = t["anchor"].shape[0], t["vis_idx"].numel(), ctx.n
        cfg, inp, cp = _structs(ctx.flags, Na, Nv, t, prm)

        def og(g, cols):
            return torch.zeros(n, cols, dtype=torch.float32, device=dev) if g is None else g.contiguous().float()
        gx, gc, go, gs, gr = og(g_xyz, 3), og(g_color, 3), og(g_opacity, 1), og(g_scaling, 3), og(g_rot, 4)
        ogr = OutGrads(ptr(gx), ptr(gc), ptr(go), ptr(gs), ptr(gr))
        # rows of invisible anchors must read as zero: ONE zero-filled flat buffer, sliced into the four gradient tensors
        shapes = [t["feat"].shape, t["anchor"].shape, t["offset"].shape, t["scaling"].shape]      # feat first: it must be 16-B aligned
        sizes = [int(torch.Size(sh).numel()) for sh in shapes]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        d_feat, d_anchor, d_offset, d_scaling = [p_.view(sh) for p_, sh in zip(torch.split(flat, sizes), shapes)]
        gp = {nm: (None if prm[nm] is None else torch.empty_like(prm[nm])) for nm in PARAM_NAMES}
        ig = InGrads(ptr(d_anchor), ptr(d_feat), ptr(d_offset), ptr(d_scaling), Params(*[ptr(gp[nm]) for nm in PARAM_NAMES]))
        scratch = torch.empty(L.gsd_backward_scratch_bytes(C.byref(cfg)), dtype=torch.uint8, device=dev)
        # parameters are unchanged between forward and backward of one graph, so the forward's repacked weights are reused
        check(L.gsd_backward(C.byref(cfg), C.byref(inp), C.byref(cp), ptr(nop), ptr(row_offset), n, C.byref(ogr), C.byref(ig),
                             ptr(fwd_scratch), ptr(scratch), scratch.numel(), stream_ptr(dev)), "decode backward")
        return (None, None, None, None, None, d_anchor, d_feat, d_offset, d_scaling, *[gp[nm] for nm in PARAM_NAMES])


def _head(mlp):
    """nn.Sequential(Linear, ReLU, Linear[, act]) or a (W1, b1, W2, b2) tuple -> the four tensors."""
    if isinstance(mlp, (tuple, list)):
        return tuple(mlp)
    return mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias


def feature_bank_blend(anchor, feat, vis_idx, campos, mlp_feature_bank):
    """The `use_feat_bank` branch of generate_neural_gaussians (scaffold_scene.py:45-56 / octree_scene.py:45-56): per visible anchor
    a softmax over three resolutions, w = softmax(Linear(32,3)(relu(Linear(4,32)([view, dist])))), and
        feat' = feat[::4].repeat(4) * w0 + feat[::2].repeat(2) * w1 + feat * w2.
    Plain differentiable torch on the Nv visible rows (four tiny ops; the branch is off by default in the reference,
    scaffold_gaussian.py:40), written back into an (Na, 32) array so that the fused kernel's gather by vis_idx is unchanged."""
    W1, b1, W2, b2 = _head(mlp_feature_bank)
    vi = vis_idx.long()
    a = anchor.index_select(0, vi); f = feat.index_select(0, vi)
    ob = a - campos
    dist = ob.norm(dim=1, keepdim=True)
    x = torch.cat([ob / dist, dist], dim=1)
    w = torch.softmax(torch.relu(x @ W1.t() + b1) @ W2.t() + b2, dim=1)
    fb = f[:, ::4].repeat(1, 4) * w[:, :1] + f[:, ::2].repeat(1, 2) * w[:, 1:2] + f * w[:, 2:]
    return torch.zeros_like(feat).index_copy(0, vi, fb)


def neural_gaussians(anchor, feat, offset, scaling, mlp_opacity, mlp_cov, mlp_color, campos, visible_mask=None, vis_idx=None,
                     appearance=None, level=None, opacity_scale=None, add_opacity_dist=False, add_cov_dist=False, add_color_dist=False,
                     use_feat_bank=False, mlp_feature_bank=None, padded=False, static_rows=False, deferred=False):
    """-> (xyz, color, opacity, scaling, rot, neural_opacity, mask), the `is_training=True` tuple of the reference.

    anchor (Na,3), feat (Na,32), offset (Na,k,3), scaling (Na,6) = get_scaling; `appearance` = embedding_appearance row of this camera
    ((A,) tensor, keeps its autograd link to the embedding table); `level` (Na,) or (Na,1) when add_level; `opacity_scale` (Na,) = the
    Octree progressive ratio with prog[~transition_mask] = 1.  `visible_mask` (bool, Na) or `vis_idx` (int32 indices) selects the anchors.
    `use_feat_bank=True` (+ `mlp_feature_bank`) = the reference's view-adaptive feature branch, see feature_bank_blend.
    `padded=True` (or a `vis_idx` from compact_visible(mask, padded=True)): no host synchronisation for the visible-anchor count; the returned
    `neural_opacity` / `mask` then have Na * k rows (zeros behind the visible anchors' rows) -- what training_stats_ accepts as is.
    `static_rows=True` (round 3; implies a padded visible list): NO host synchronisation at all and STATIC output shapes -- the form a HIP graph can
    record.  The five Gaussian tensors keep all Nv * k rows: the P emitted Gaussians first, the rest parked at the camera centre with zero opacity
    (every rasterizer of this library culls them: radii 0, no tile instance, zero gradients), and an eighth value `count` (int32 device tensor,
    shape (1,)) = P is returned for consumers that average over the Gaussians (the reference's scaling loss: use sum() / count).  The parked rows
    sit at view depth 0, i.e. they FAIL the frustum test by construction: rasterize them with `prefiltered=False` (what every scene of the reference
    passes, e.g. scaffold_scene.py:101) -- with `prefiltered=True` the rasterizer reports them as the reference does ("Point is filtered ...").
    `deferred=True` (round 4): the kernels are enqueued and a PendingDecode is returned instead of the tuple; its finish() waits for the count and
    returns the reference-shaped tuple.  A caller that renders two cameras per iteration (PGSR after step 7000) starts the second camera's decode
    before finishing the first: the host's wait for the first count is covered by the second decode's kernels."""
    if use_feat_bank and mlp_feature_bank is None:
        raise RuntimeError("gsrast.decode: use_feat_bank=True needs mlp_feature_bank (get_featurebank_mlp of the gaussian model)")
    if feat.shape[1] != 32:
        raise NotImplementedError("gsrast.decode: feat_dim must be 32")
    Na, k = offset.shape[0], offset.shape[1]
    if vis_idx is None:
        vis_idx = torch.arange(Na, dtype=torch.int32, device=anchor.device) if visible_mask is None else compact_visible(visible_mask, padded)
    if use_feat_bank:
        if padded or (vis_idx.numel() and bool((vis_idx[-1:] < 0).any())):
            raise RuntimeError("gsrast.decode: use_feat_bank needs the exact visible list (padded=False)")
        feat = feature_bank_blend(anchor, feat, vis_idx, campos, mlp_feature_bank)
    heads = _head(mlp_opacity) + _head(mlp_cov) + _head(mlp_color)
    A = 0 if appearance is None else appearance.numel()
    lvl = None if level is None else level.reshape(-1)
    osc = None if opacity_scale is None else opacity_scale.reshape(-1)
    flags = (int(k), int(A), bool(add_opacity_dist), bool(add_cov_dist), bool(add_color_dist), lvl is not None, bool(static_rows))
    app = None if appearance is None else appearance.reshape(-1)
    if deferred:
        if static_rows:
            raise RuntimeError("gsrast.decode: deferred=True returns the reference-shaped tuple; static_rows=True has no count to wait for")
        args = (vis_idx, campos, lvl, osc, anchor, feat, offset, scaling, *heads, app)
        with torch.no_grad():
            t, prm = _decode_inputs(vis_idx, campos, lvl, osc, anchor, feat, offset, scaling, (*heads, app))
            launched = _decode_launch(flags[:6], t, prm, "deferred")
        return PendingDecode(flags, args, launched)
    return _NeuralDecode.apply(flags, vis_idx, campos, lvl, osc, anchor, feat, offset, scaling, *heads, app)


def training_stats_(opacity_accum, anchor_demon, offset_gradient_accum, offset_denom, viewspace_grad, neural_opacity, update_filter,
                    offset_selection_mask, anchor_visible_mask=None, vis_idx=None):
    """In-place `ScaffoldGaussian.training_statis` (gssr/gaussian/scaffold_gaussian.py:488-508; same argument meaning: `viewspace_grad` =
    viewspace_point_tensor.grad (P, >=2), `neural_opacity` = the decode's pre-gate opacity (Nv*k), `update_filter` = visibility_filter (P,),
    `offset_selection_mask` = the decode's mask (Nv*k), `anchor_visible_mask` (Na,) bool -- or the `vis_idx` the decode already computed).
    The four accumulators are the model's float32 buffers ((Na,1), (Na,1), (Na*k,1), (Na*k,1)); updated without any host synchronisation."""
    L = _lib()
    with torch.no_grad():
        if vis_idx is None:
            vis_idx = compact_visible(anchor_visible_mask)
        Nv = int(vis_idx.numel())
        if Nv == 0:
            return
        sel = offset_selection_mask.reshape(-1)
        k = sel.numel() // Nv
        if sel.numel() != Nv * k or neural_opacity.numel() != Nv * k:
            raise RuntimeError("training_stats_: neural_opacity / offset_selection_mask must have Nv * n_offsets entries")
        g = dev_f32(viewspace_grad, "viewspace_grad")
        P = g.shape[0]
        upd = update_filter.reshape(-1)
        if upd.numel() != P:
            raise RuntimeError("training_stats_: update_filter must have one entry per generated Gaussian")
        for t, n, cnt in ((opacity_accum, "opacity_accum", None), (anchor_demon, "anchor_demon", None),
                          (offset_gradient_accum, "offset_gradient_accum", None), (offset_denom, "offset_denom", None)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise RuntimeError(f"training_stats_: {n} must be a contiguous float32 HIP tensor")
        if offset_gradient_accum.numel() != opacity_accum.numel() * k or offset_denom.numel() != offset_gradient_accum.numel():
            raise RuntimeError("training_stats_: accumulator sizes do not match (Na, Na*k)")
        no = dev_f32(neural_opacity.reshape(-1), "neural_opacity")
        sel8 = sel.contiguous().view(torch.uint8) if sel.dtype == torch.bool else sel.to(torch.uint8).contiguous()
        upd8 = upd.contiguous().view(torch.uint8) if upd.dtype == torch.bool else upd.to(torch.uint8).contiguous()
        vi = vis_idx.to(torch.int32).contiguous()
        dev = g.device
        scratch = torch.empty(max(L.gsd_training_stats_scratch_bytes(Nv), 8), dtype=torch.uint8, device=dev)
        check(L.gsd_training_stats(Nv, k, ptr(vi), ptr(no), ptr(sel8), ptr(upd8), ptr(g), int(g.shape[1]), ptr(opacity_accum), ptr(anchor_demon),
                                   ptr(offset_gradient_accum), ptr(offset_denom), ptr(scratch), scratch.numel(), stream_ptr(dev)), "training_stats")
