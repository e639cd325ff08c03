"""HIP TSDF voxel integration: the per-frame update of GaussianExtractor.extract_mesh_unbounded's
compute_unbounded_tsdf (gssr/utils/mesh_utils.py:195-246) as one streaming kernel."""
import torch

from . import last_error, lib, check, ptr, stream_ptr, dev_f32, TsdfSparse


def tsdf_integrate_(points, full_proj_transform, depthmap, rgbmap, sdf_trunc, tsdfs, rgbs, weights):
    """In place.  points [V,3]; depthmap [1,H,W] or [H,W]; rgbmap [3,H,W]; sdf_trunc float or tensor [V];
    tsdfs [V], rgbs [V,3], weights [V] float32 CUDA tensors (initial values: 1, 0, 1 as in the reference)."""
    pts = dev_f32(points, "points", allow_empty=False)
    F = dev_f32(full_proj_transform, "full_proj_transform", allow_empty=False)
    d = dev_f32(depthmap, "depthmap", allow_empty=False)
    c = dev_f32(rgbmap, "rgbmap", allow_empty=False)
    H, W = int(d.shape[-2]), int(d.shape[-1])
    for t, n in ((tsdfs, "tsdfs"), (rgbs, "rgbs"), (weights, "weights")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"{n} must be a contiguous float32 CUDA tensor")
    tp = None
    st = 0.0
    if isinstance(sdf_trunc, torch.Tensor):
        tp = dev_f32(sdf_trunc, "sdf_trunc", allow_empty=False)
    else:
        st = float(sdf_trunc)
    scratch = torch.empty((H * W, 4), dtype=torch.float32, device=pts.device)      # interleaved (r,g,b,d) texels
    check(lib().gsr_tsdf_integrate(int(pts.shape[0]), ptr(pts), ptr(F), W, H, ptr(d), ptr(c), st, ptr(tp), ptr(tsdfs),
                                   ptr(weights), ptr(rgbs), ptr(scratch), stream_ptr(pts.device)), "tsdf_integrate")
    return tsdfs, rgbs, weights


class DenseTSDFVolume:
    """Dense-grid stand-in for o3d.pipelines.integration.ScalableTSDFVolume as GS-SR uses it
    (gssr/utils/mesh_utils.py:154-178: voxel_length, sdf_trunc, RGB8 colours, depth_scale=1, depth_trunc).
    Open3D is not vendored in the reference: the voxel update follows Open3D's published UniformTSDFVolume algorithm and
    its parity is UNPINNED.  State: tsdf, weight [nx,ny,nz], color [nx,ny,nz,3] float32 on the HIP device.

    Multi-GPU (extract_mesh_split.py fuses the frames of all tiles into one volume): every rank integrates its own
    frames into its own volume, then merge_() all-reduces the two associative accumulators (sum w*tsdf, sum w) over
    RCCL -- the weighted average is order-independent up to fp32 rounding."""

    def __init__(self, origin, voxel_length, dims, sdf_trunc, device="cuda"):
        self.origin = [float(v) for v in origin]
        self.voxel_length = float(voxel_length)
        self.dims = tuple(int(d) for d in dims)
        self.sdf_trunc = float(sdf_trunc)
        self.device = torch.device(device)
        self.tsdf = torch.zeros(self.dims, dtype=torch.float32, device=self.device)
        self.weight = torch.zeros(self.dims, dtype=torch.float32, device=self.device)
        self.color = torch.zeros(self.dims + (3,), dtype=torch.float32, device=self.device)

    def integrate(self, rgb, depth, fx, fy, cx, cy, extrinsic, depth_trunc=float("inf"), quantize_rgb8=True):
        """rgb [3,H,W] in [0,1], depth [1,H,W] or [H,W], extrinsic 4x4 world->camera (row-major, Open3D convention)."""
        import ctypes as C
        d = dev_f32(depth, "depth", allow_empty=False)
        c = dev_f32(rgb, "rgb", allow_empty=False)
        if quantize_rgb8:        # mesh_utils.py:170 converts colours to uint8 before fusion
            c = (torch.clamp(c, 0.0, 1.0) * 255).to(torch.uint8).to(torch.float32).contiguous()
        else:                    # same 0..255 scale without the rounding, so that volumes built either way can be merged
            c = (torch.clamp(c, 0.0, 1.0) * 255).contiguous()
        H, W = int(d.shape[-2]), int(d.shape[-1])
        E = (C.c_float * 16)(*[float(v) for v in torch.as_tensor(extrinsic, dtype=torch.float32).reshape(-1).tolist()])
        o = (C.c_float * 3)(*self.origin)
        check(lib().gsr_tsdf_integrate_dense(self.dims[0], self.dims[1], self.dims[2], o, self.voxel_length, self.sdf_trunc,
                                             float(min(depth_trunc, 3.0e38)), W, H, ptr(d), ptr(c), float(fx), float(fy), float(cx),
                                             float(cy), E, ptr(self.tsdf), ptr(self.weight), ptr(self.color),
                                             stream_ptr(self.device)), "tsdf_integrate_dense")
        return self

    def merge_(self, group=None):
        """All-reduce (sum) of the weighted accumulators across ranks; afterwards every rank holds the fused volume."""
        merge_volumes_(self.tsdf, self.weight, self.color, group)
        return self


def merge_volumes_(tsdf, weight, color, group=None):
    """tsdf <- sum_r(w_r*tsdf_r)/sum_r(w_r), color likewise, weight <- sum_r(w_r); in place, any device/backend."""
    import torch.distributed as dist
    acc = tsdf * weight
    cacc = color * weight.unsqueeze(-1)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(cacc, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(weight, op=dist.ReduceOp.SUM, group=group)
    nz = weight > 0
    tsdf.copy_(torch.where(nz, acc / weight.clamp_min(1e-30), torch.zeros_like(acc)))
    color.copy_(torch.where(nz.unsqueeze(-1), cacc / weight.clamp_min(1e-30).unsqueeze(-1), torch.zeros_like(cacc)))
    return tsdf, weight, color


def brick_to_xmajor(planes):
    """[..., 4096] unit planes in the storage order of include/gsrast.h (ABI 8: float index 4 g + (z & 3),
    g = (x>>2)<<8 | (y>>2)<<6 | (z>>2)<<4 | ((x>>1)&1)<<3 | ((y>>1)&1)<<2 | (x&1)<<1 | (y&1)) -> [..., 16, 16, 16] indexed [x, y, z].  Any device."""
    lead = tuple(planes.shape[:-1])
    k = len(lead)
    # storage index bits, high to low: bx by bz (2 each) | x1 y1 x0 y0 | z (2)
    v = planes.reshape(lead + (4, 4, 4, 2, 2, 2, 2, 4))
    return v.permute(*range(k), k + 0, k + 3, k + 5, k + 1, k + 4, k + 6, k + 2, k + 7).reshape(lead + (16, 16, 16))


class ScalableTSDFVolume:
    """Block-sparse TSDF volume with the call shape of o3d.pipelines.integration.ScalableTSDFVolume as GS-SR drives it
    (gssr/utils/mesh_utils.py:154-178: `ScalableTSDFVolume(voxel_length, sdf_trunc, RGB8)`, `integrate(rgbd, intrinsic, extrinsic)` with
    depth_scale 1 and depth_trunc; extract_mesh.py:125-128 picks voxel_length = depth_trunc / 1024 and sdf_trunc = 5 voxels -- a dense
    grid of that resolution would be >= 1024^3 voxels, here only the 16^3 units near the observed surface exist).
    Open3D is not part of the reference tree: PARITY UNPINNED (algorithm restated from Open3D 0.18's published sources, see
    csrc/gsr_tsdf_sparse.hip); tests pin the HIP volume against a plain-C restatement on the CPU and, unit by unit, against
    DenseTSDFVolume.

    capacity_units is the INITIAL number of 16^3 units (80 KB each: tsdf + weight + 3 colour floats per voxel; rounded up to a power of two); a frame
    that needs more doubles the pool -- one more chunk of records, no voxel copied -- unless auto_grow=False, in which case it raises.  With auto_grow the pool
    also grows ahead of need (once more than half of it is allocated), so that a render -> integrate loop rarely sees a refused frame.
    Colours are stored on the 0..255 scale whether or not they are quantised to integers first (quantize_rgb8).
    The pools hold a unit in brick order and only the 16-byte groups a frame or merge has written (`mask`, ABI 8): read them through `units()` /
    `to_dense()`, which hand out plain x-major arrays.
    Multi-GPU (extract_mesh_split.py:54-128): every rank integrates its own tile's frames, `merge_()` fuses the volumes of all ranks
    (weighted running averages are associative), `merge_from(other)` fuses two volumes on one device."""

    RES = 16
    TSDF_NO_SYNC = 1
    MAX_IN_FLIGHT = 8             # deferred frames whose status words nobody has looked at yet

    def __init__(self, voxel_length, sdf_trunc, capacity_units=16384, device="cuda", depth_sampling_stride=4, auto_grow=True):
        self.voxel_length = float(voxel_length)
        self.sdf_trunc = float(sdf_trunc)
        self.auto_grow = bool(auto_grow)
        if 2.0 * self.sdf_trunc > 3.0 * self.RES * self.voxel_length:
            raise RuntimeError(f"ScalableTSDFVolume: sdf_trunc {self.sdf_trunc} exceeds 1.5 units = {1.5 * self.RES * self.voxel_length} (24 voxels)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ScalableTSDFVolume lives on a HIP device; there is no CPU path")
        self.stride = int(depth_sampling_stride)
        self._alloc(1 << max(4, (int(capacity_units) - 1).bit_length()))              # chunk addressing wants powers of two
        self.frame = 0
        self.last_touched = 0
        self._allocated = 0           # units allocated as of the last frame whose status words have been read
        self._tex = None              # texel scratch of the last frame size (frames of one stream run in order: one buffer serves every frame in flight)
        self._queue = []              # frames enqueued with defer=True whose status words have not been looked at yet, oldest first
        self._status = None           # pinned int32[MAX_IN_FLIGHT, 4]

    @property
    def _pending(self):
        """The newest frame still in flight (None when the volume is settled)."""
        return self._queue[-1] if self._queue else None

    def _alloc(self, cap, chunks=None):
        """The volume's arrays for `cap` units (a power of two).  Voxels live in unit records of 5 x 4096 floats (planes tsdf, weight, r, g, b) held in chunks
        of doubling size -- `chunks` keeps the existing ones, the rest is allocated.  Nothing is initialised, neither the records (80 KB per unit) nor the
        written-group words: a unit whose stamp is 0 has never been written, and a clear mask bit means "zero" whatever the record holds -- a 131 072-unit
        volume used to start with a 10.7 GB memset."""
        d = self.device
        self.cap = int(cap)
        self.log2 = max(4, (2 * self.cap - 1).bit_length())
        self.keys = torch.full((1 << self.log2,), -1, dtype=torch.int64, device=d)
        self.slot = torch.zeros((1 << self.log2,), dtype=torch.int32, device=d)
        self.coord = torch.zeros((self.cap, 3), dtype=torch.int32, device=d)
        self.stamp = torch.zeros((self.cap,), dtype=torch.int32, device=d)
        self.list = torch.zeros((self.cap,), dtype=torch.int32, device=d)
        self.counters = torch.zeros((4,), dtype=torch.int32, device=d)
        self.mask = torch.empty((self.cap, 16), dtype=torch.int64, device=d)           # written-group bits (ABI 8)
        self.chunks = list(chunks or [])
        have = sum(int(c.shape[0]) for c in self.chunks)
        while have < self.cap:
            k = self.cap if not self.chunks else have                                  # chunk 0: the initial capacity; every further chunk doubles the pool
            if len(self.chunks) >= TsdfSparse.MAX_CHUNKS:
                raise RuntimeError("ScalableTSDFVolume: chunk table full")
            self.chunks.append(torch.empty((k, 5, self.RES ** 3), dtype=torch.float32, device=d))
            have += k
        self.chunk0_log2 = int(self.chunks[0].shape[0]).bit_length() - 1

    def _struct(self):
        import ctypes as C
        arr = (C.c_void_p * TsdfSparse.MAX_CHUNKS)(*[c.data_ptr() for c in self.chunks])
        return TsdfSparse(ptr(self.keys), ptr(self.slot), ptr(self.coord), ptr(self.stamp), ptr(self.list), ptr(self.counters), ptr(self.mask), arr,
                          self.chunk0_log2, len(self.chunks), self.log2, self.cap, self.voxel_length, self.sdf_trunc)

    def records(self, n=None):
        """[n, 5, 4096] the raw unit records of units [0, n) (planes tsdf, weight, r, g, b in brick order; a view when they live in one chunk, else a
        concatenation): test / export helper -- unwritten groups hold whatever the memory held unless the volume has been materialised."""
        n = self.num_units if n is None else int(n)
        parts, s = [], 0
        for c in self.chunks:
            k = min(int(c.shape[0]), n - s)
            if k <= 0:
                break
            parts.append(c[:k]); s += k
        return parts[0] if len(parts) == 1 else torch.cat(parts, 0) if parts else self.chunks[0][:0]

    def integrate(self, rgb, depth, fx, fy, cx, cy, extrinsic, depth_trunc=float("inf"), quantize_rgb8=True, defer=False):
        """rgb [3,H,W] in [0,1], depth [1,H,W] or [H,W] (0 = invalid, as mesh_utils.py:165-166 writes for masked pixels),
        extrinsic 4x4 world->camera (Open3D convention).  Colours are put on the 0..255 scale on the device (quantize_rgb8: through the uint8
        truncation of mesh_utils.py:170).
        defer=True: the frame is only ENQUEUED (no host synchronisation at all: texels, touch, stamp and the voxel pass run back to back behind whatever
        produced rgb / depth); up to MAX_IN_FLIGHT frames wait that way.  Their outcome -- pool exhausted, sample out of range -- is looked at when the
        queue is full, by finish() or by any read of the volume, which then grows the pool and runs the refused frame AND every frame enqueued after it
        again, in order (a refused frame has integrated nothing, and neither has any frame behind it).  The volume keeps a reference to a deferred frame's
        rgb / depth until then: the caller must not overwrite those tensors before finish() (pass clones if its render buffers are reused)."""
        import ctypes as C
        d = dev_f32(depth, "depth", allow_empty=False)
        c = dev_f32(rgb, "rgb", allow_empty=False)
        H, W = int(d.shape[-2]), int(d.shape[-1])
        if tuple(c.shape[-2:]) != (H, W) or c.numel() != 3 * H * W:
            raise RuntimeError("ScalableTSDFVolume.integrate: rgb must be [3,H,W] with the depth map's H, W")
        E = torch.as_tensor(extrinsic, dtype=torch.float64).reshape(4, 4).cpu()
        Pm = torch.linalg.inv(E)
        Ea = (C.c_float * 12)(*[float(v) for v in E[:3].reshape(-1).tolist()])
        Pa = (C.c_float * 12)(*[float(v) for v in Pm[:3].reshape(-1).tolist()])
        if not defer or len(self._queue) >= self.MAX_IN_FLIGHT:
            self.finish()
        elif self._queue and self._queue[0]["event"].query():
            self._settle_done()
        if self.auto_grow and 2 * self._allocated > self.cap and self.cap < (1 << 27):
            # grow ahead of need: with chunked records that costs an allocation and a re-key of the (small) table, not a copy -- a refused frame costs its
            # touch passes twice and stalls every frame queued behind it
            self.finish()
            while 2 * self._allocated > self.cap and self.cap < (1 << 27):
                self._grow()
        if self._status is None:
            self._status = torch.zeros((self.MAX_IN_FLIGHT, 4), dtype=torch.int32).pin_memory()
        frame = dict(d=d, c=c, W=W, H=H, intr=(float(fx), float(fy), float(cx), float(cy)), Ea=Ea, Pa=Pa, dt=float(min(depth_trunc, 3.0e38)),
                     quant=2 if quantize_rgb8 else 1)
        self._enqueue(frame, sync=not defer)
        if defer:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            frame["event"] = ev
            self._queue.append(frame)
        else:
            self._resolve([frame])
        return self

    def _enqueue(self, f, sync):
        import ctypes as C
        self.frame += 1
        used = {q["slot"] for q in self._queue}
        f["slot"] = next(i for i in range(self.MAX_IN_FLIGHT + 1) if i not in used) % self.MAX_IN_FLIGHT
        tex = self._tex
        if tex is None or tex.numel() != 4 * f["H"] * f["W"]:
            tex = self._tex = torch.empty((f["H"] * f["W"], 4), dtype=torch.float32, device=self.device)
        st = self._struct()
        with torch.cuda.device(self.device):
            rc = lib().gsr_tsdf_sparse_integrate2(C.byref(st), f["W"], f["H"], ptr(f["d"]), ptr(f["c"]), f["quant"], *f["intr"], f["Ea"], f["Pa"], f["dt"],
                                                  self.stride, self.frame, ptr(tex), C.c_void_p(self._status[f["slot"]].data_ptr()),
                                                  0 if sync else self.TSDF_NO_SYNC, stream_ptr(self.device))
        f["rc"] = rc

    def _frame_rc(self, f):
        import ctypes as C
        rc = f["rc"]
        if rc == 0 and "event" in f:
            st = self._struct()
            with torch.cuda.device(self.device):
                rc = lib().gsr_tsdf_sparse_status(C.byref(st), C.c_void_p(self._status[f["slot"]].data_ptr()), stream_ptr(self.device))
        return rc

    def _resolve(self, frames):
        """Looks at the outcome of enqueued frames (oldest first) whose status words are on the host.  "Capacity exhausted": nothing of that frame nor of the
        frames behind it has been integrated -- the pool is grown and they run again, in order.  Anything else the library reports is raised (the frames
        behind the failing one are dropped with it: they integrated nothing either)."""
        frames = list(frames)
        rerun = False
        while frames:
            f = frames[0]
            if rerun:
                f.pop("event", None)
                self._enqueue(f, sync=True)
            rc = self._frame_rc(f)
            if rc != 0 and self.auto_grow and "capacity exhausted" in last_error() and self.cap < (1 << 27):
                self._grow()
                rerun = True
                continue
            if rc != 0:
                msg = last_error()
                self._after_failure()
                raise RuntimeError(f"tsdf_sparse_integrate: {msg}")
            self.last_touched = int(self._status[f["slot"], 1]); self._allocated = int(self._status[f["slot"], 0])
            frames.pop(0)

    def _settle_done(self):
        """Resolves the frames at the head of the queue whose event has already fired (no waiting)."""
        while self._queue and self._queue[0]["event"].query():
            f = self._queue[0]
            if self._frame_rc(f) != 0:
                return self.finish()
            self.last_touched = int(self._status[f["slot"], 1]); self._allocated = int(self._status[f["slot"], 0])
            self._queue.pop(0)

    def finish(self):
        """Waits for the frames enqueued with defer=True (if any) and handles their outcome.  Every method that reads the volume calls it."""
        q, self._queue = self._queue, []
        if q:
            q[-1]["event"].synchronize()
            self._resolve(q)
        return self

    def _after_failure(self):
        """A frame or merge failed for good (auto_grow off, or a sample out of range): the unit counter goes back into the pool (keys the failed call put
        into the table without a pool slot stay harmless: they resolve to slot -1) and the failure flag is cleared, so that the volume stays usable and
        `num_units` never exceeds the capacity.  Units the failed call allocated but never wrote keep stamp 0: units() shows them as empty."""
        n = min(int(self.counters[0].item()), self.cap)
        self.counters[0] = n
        self.counters[2] = 0

    def _grow(self):
        """Doubles the unit pool: ONE MORE CHUNK of records (no voxel is copied, the volume never holds old and new pools side by side), larger coord / stamp /
        list / mask arrays with the allocated units' entries copied (140 B per unit), the hash table re-keyed with every unit in its old slot."""
        import ctypes as C
        n = min(int(self.counters[0].item()), self.cap)
        old = (self.coord, self.stamp, self.mask)
        self._alloc(2 * self.cap, chunks=self.chunks)
        for dst, src in zip((self.coord, self.stamp, self.mask), old):
            dst[:n].copy_(src[:n])
        self.counters[0] = n
        st = self._struct()
        with torch.cuda.device(self.device):
            check(lib().gsr_tsdf_sparse_rehash(C.byref(st), n, stream_ptr(self.device)), "tsdf_sparse_rehash")

    @property
    def num_units(self):
        self.finish()
        return min(int(self.counters[0].item()), self.cap)

    def weight_sum(self):
        """Sum of the weights of every written voxel (float64) WITHOUT materialising the volume -- every update of a voxel adds exactly 1, so the difference
        across a frame is the number of voxels the frame updated (the benchmarks' algorithmic byte count)."""
        n = self.num_units
        if n == 0:
            return 0.0
        tot, s0 = 0.0, 0
        for ch in self.chunks:
            for s in range(s0, min(n, s0 + int(ch.shape[0])), 16384):
                e = min(n, s0 + int(ch.shape[0]), s + 16384)
                bits = ((self.mask[s:e].view(e - s, 16, 1) >> torch.arange(64, device=self.device).view(1, 1, 64)) & 1).bool().view(e - s, 1024)
                bits &= (self.stamp[s:e] != 0).view(-1, 1)
                w = ch[s - s0:e - s0, 1].view(e - s, 1024, 4)
                tot += float(torch.where(bits.unsqueeze(-1), w, torch.zeros((), device=self.device)).double().sum())
            s0 += int(ch.shape[0])
        return tot

    def units(self):
        """-> (coords [n,3] int32, tsdf [n,16,16,16], weight [n,16,16,16], color [n,16,16,16,3]) of the allocated units as plain arrays
        (voxel index x-major, z fastest, like Open3D's UniformTSDFVolume::IndexOf).  Groups no frame ever wrote are zero-filled in the pools first
        (gsr_tsdf_sparse_materialize); the arrays are re-ordered COPIES of the brick-ordered pools."""
        import ctypes as C
        n = self.num_units
        st = self._struct()
        with torch.cuda.device(self.device):
            check(lib().gsr_tsdf_sparse_materialize(C.byref(st), n, stream_ptr(self.device)), "tsdf_sparse_materialize")
        # storage index bits, high to low: bx by bz (2 each) | x1 y1 x0 y0 | z (2)
        rec = brick_to_xmajor(self.records(n))                      # [n, 5, 16, 16, 16]
        return self.coord[:n], rec[:, 0], rec[:, 1], rec[:, 2:5].permute(0, 2, 3, 4, 1)

    def merge_units_(self, coords, tsdf, weight, color, assume_unique=False):
        """self <- weighted merge with the given units (plain arrays shaped like `units()`, on this device).  The merge kernel runs one workgroup per
        listed unit, so a coordinate must not appear twice in one list: unless `assume_unique` the list is first fused with itself
        (merge_unit_lists: sort-unique + index_add)."""
        import ctypes as C
        self.finish()
        n = int(coords.shape[0])
        if n == 0:
            return self
        if not assume_unique:
            V = self.RES ** 3
            coords, tsdf, weight, color = merge_unit_lists(coords, tsdf.reshape(n, V), weight.reshape(n, V), color.reshape(n, V, 3))
            n = int(coords.shape[0])
        co = coords.to(torch.int32).contiguous()
        t, w, c = (x.to(torch.float32).contiguous() for x in (tsdf, weight, color))
        self._merge_call(lambda st: lib().gsr_tsdf_sparse_merge(C.byref(st), n, ptr(co), ptr(t), ptr(w), ptr(c), stream_ptr(self.device)), "tsdf_sparse_merge")
        return self

    def _merge_call(self, call, what):
        """One merge through the C ABI; "capacity exhausted" grows the pool and runs it again (the units the refused call allocated keep stamp 0 and are
        found again), anything else leaves the volume usable and raises."""
        while True:
            st = self._struct()
            with torch.cuda.device(self.device):
                rc = call(st)
            if rc != 0 and self.auto_grow and "capacity exhausted" in last_error() and self.cap < (1 << 27):
                self._grow()
                continue
            if rc != 0:
                msg = last_error()
                self._after_failure()
                raise RuntimeError(f"{what}: {msg}")
            self._allocated = min(int(self.counters[0].item()), self.cap)
            return

    def merge_from(self, other):
        """Fuses another volume (same voxel_length / sdf_trunc) into this one, e.g. the volumes of two tiles.  On one device the other volume's pools are
        read where they lie (gsr_tsdf_sparse_merge_volume: no export, no re-ordering, unwritten groups never touched)."""
        import ctypes as C
        if abs(other.voxel_length - self.voxel_length) > 0 or abs(other.sdf_trunc - self.sdf_trunc) > 0:
            raise RuntimeError("merge_from: volumes must share voxel_length and sdf_trunc")
        self.finish()
        n = other.num_units
        if other.device == self.device:
            if n:
                so = other._struct()
                self._merge_call(lambda st: lib().gsr_tsdf_sparse_merge_volume(C.byref(st), C.byref(so), n, stream_ptr(self.device)), "tsdf_sparse_merge_volume")
            return self
        co, t, w, c = other.units()
        return self.merge_units_(co.to(self.device), t.to(self.device), w.to(self.device), c.to(self.device), assume_unique=True)

    def merge_(self, group=None):
        """Fuses the volumes of all ranks; afterwards every rank holds the same fused volume (unit numbering may differ between ranks).
        One all_gather of the unit counts, then one padded all_gather per array: the payload is the allocated units only."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        co, t, w, c = self.units()
        merged = merge_unit_lists(*gather_unit_lists(co, t.reshape(len(co), -1), w.reshape(len(co), -1), c.reshape(len(co), -1, 3), group))
        fresh = ScalableTSDFVolume(self.voxel_length, self.sdf_trunc, max(self.cap, int(merged[0].shape[0])), self.device, self.stride, self.auto_grow)
        fresh.merge_units_(*merged, assume_unique=True)
        fresh.frame = self.frame
        self.__dict__.update(fresh.__dict__)
        return self

    def to_dense(self, origin_unit, dims_units):
        """Dense (tsdf, weight, color) arrays over a box of units [origin_unit, origin_unit + dims_units): test / export helper."""
        R = self.RES
        nx, ny, nz = (int(v) for v in dims_units)
        org = torch.tensor([int(v) for v in origin_unit], dtype=torch.int32, device=self.device)
        co, t, w, c = self.units()
        rel = co - org
        ins = ((rel >= 0) & (rel < torch.tensor([nx, ny, nz], dtype=torch.int32, device=self.device))).all(dim=1)
        idx = rel[ins].long()
        # (nx, ny, nz, R, R, R) unit-major scratch: every selected unit is one indexed assignment, then the axes are interleaved
        T = torch.zeros((nx, ny, nz, R, R, R), dtype=torch.float32, device=self.device)
        Wt = torch.zeros_like(T); Cc = torch.zeros((nx, ny, nz, R, R, R, 3), dtype=torch.float32, device=self.device)
        T[idx[:, 0], idx[:, 1], idx[:, 2]] = t[ins]; Wt[idx[:, 0], idx[:, 1], idx[:, 2]] = w[ins]; Cc[idx[:, 0], idx[:, 1], idx[:, 2]] = c[ins]
        dense = lambda a: a.permute(0, 3, 1, 4, 2, 5, *range(6, a.dim())).reshape(nx * R, ny * R, nz * R, *a.shape[6:])
        return dense(T), dense(Wt), dense(Cc)


def gather_unit_lists(coords, tsdf, weight, color, group=None):
    """all_gather of per-rank unit lists (any device / backend): -> concatenated (coords, tsdf, weight, color) of every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([coords.shape[0]], dtype=torch.int64, device=coords.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)

    def gather(x):
        pad = torch.zeros((m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[: x.shape[0]] = x
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(out, pad, group=group)
        return torch.cat([o[:k] for o, k in zip(out, counts)], dim=0)
    return gather(coords.contiguous()), gather(tsdf.contiguous()), gather(weight.contiguous()), gather(color.contiguous())


def merge_unit_lists(coords, tsdf, weight, color):
    """Weighted merge of a concatenated unit list with repeated coordinates (pure torch, any device):
    tsdf = sum(w * tsdf) / sum(w), colour likewise, weight = sum(w).  -> (unique coords [m,3], tsdf [m,V], weight [m,V], color [m,V,3])"""
    uniq, inv = torch.unique(coords.to(torch.int64), dim=0, return_inverse=True)
    m = uniq.shape[0]
    W = torch.zeros((m,) + tuple(weight.shape[1:]), dtype=torch.float32, device=weight.device).index_add_(0, inv, weight)
    T = torch.zeros_like(W).index_add_(0, inv, tsdf * weight)
    Cc = torch.zeros((m,) + tuple(color.shape[1:]), dtype=torch.float32, device=color.device).index_add_(0, inv, color * weight.unsqueeze(-1))
    nz = W > 0
    T = torch.where(nz, T / W.clamp_min(1e-30), torch.zeros_like(T))
    Cc = torch.where(nz.unsqueeze(-1), Cc / W.clamp_min(1e-30).unsqueeze(-1), torch.zeros_like(Cc))
    return uniq.to(torch.int32), T, W, Cc
