"""GPU tests (pytest -m gpu) of the block-sparse TSDF volume (gsrast.tsdf.ScalableTSDFVolume -> gsr_tsdf_sparse_* of the C ABI):
against the CPU oracle (same allocated units, weights bit-exact), against the dense HIP volume unit by unit (bit-exact: same voxel
rule), and the multi-tile fusion of extract_mesh_split.py:54-128 (two per-tile volumes merged == all frames into one volume)."""
import numpy as np
import pytest
import torch

import oracle
import tsdf_cases

pytestmark = pytest.mark.gpu

VL, TR, DT = 0.02, 0.1, 6.0


def _poison(vol):
    """The pools are not zero-filled by the library any more (ABI 7): whatever a unit's 80 KB held before must never show.  ABI 8: the written-group words
    are not initialised either -- all-ones here, so that a kernel trusting a never-written unit's words would read the NaNs."""
    for ch in vol.chunks:
        ch.fill_(float("nan"))
    vol.mask.fill_(-1)
    return vol


def _hip_volume(frs, cap=4096, defer=False):
    from gsrast.tsdf import ScalableTSDFVolume
    vol = _poison(ScalableTSDFVolume(VL, TR, capacity_units=cap))
    for f in frs:
        vol.integrate(torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda(), f["fx"], f["fy"], f["cx"], f["cy"], f["E"], depth_trunc=DT,
                      defer=defer)
    return vol


def _oracle_units(frs):
    v = oracle.SparseTSDF(VL, TR)
    for f in frs:
        v.integrate(tsdf_cases.rgb8(f["rgb"]), f["depth"], f["fx"], f["fy"], f["cx"], f["cy"], f["E"], depth_trunc=DT)
    return v.units()


def _compare(vol, ref):
    co, t, w, c = (x.cpu().numpy() for x in vol.units())
    rco, rt, rw, rc = ref
    got = {tuple(k): i for i, k in enumerate(co.tolist())}
    want = {tuple(k): i for i, k in enumerate(rco.tolist())}
    assert set(got) == set(want), (len(got), len(want), len(set(got) ^ set(want)))       # same units opened
    nbad = ntot = 0
    for k, i in got.items():
        j = want[k]
        nbad += int((w[i] != rw[j]).sum()); ntot += w[i].size                                # v_rcp_f32 vs IEEE division at a gate
        same = w[i] == rw[j]
        # v_rcp_f32 in the projection can round a voxel into the neighbouring pixel: counted, must stay a vanishing fraction
        nbad += int((np.abs(t[i][same] - rt[j][same]) > 1e-4).sum())
        nbad += int((np.abs(c[i][same] - rc[j][same]).max(-1) > 0.05).sum())
    assert nbad <= 2e-4 * ntot, (nbad, ntot)
    return len(got)


def test_sparse_volume_matches_oracle_and_dense():
    from gsrast.tsdf import DenseTSDFVolume
    frs = tsdf_cases.frames(4)
    vol = _hip_volume(frs)
    n = _compare(vol, _oracle_units(frs))
    assert 50 < n < 4000 and vol.last_touched > 0
    # unit by unit against the dense HIP volume over the bounding box: a unit opened at frame k has exactly the dense volume's
    # contributions of frames >= k; units opened by the FIRST frame must equal the dense volume bit for bit
    first = _hip_volume(frs[:1])
    co = first.units()[0].cpu().numpy()
    lo, hi = co.min(0), co.max(0) + 1
    dense = DenseTSDFVolume((lo * np.float32(VL * 16)).tolist(), VL, tuple(int(d) * 16 for d in (hi - lo)), TR)
    f = frs[0]
    dense.integrate(torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda(), f["fx"], f["fy"], f["cx"], f["cy"], f["E"], depth_trunc=DT)
    T, Wt, C = first.to_dense(lo, hi - lo)
    alloc = torch.zeros_like(Wt, dtype=torch.bool)
    for k in co:
        x, y, z = (k - lo) * 16
        alloc[x:x + 16, y:y + 16, z:z + 16] = True
    dW = (Wt != dense.weight) & alloc
    assert dW.float().mean().item() < 1e-4                                                  # float32 voxel-centre rounding only
    ok = alloc & ~dW
    assert ((T[ok] - dense.tsdf[ok]).abs() > 2e-4).float().mean().item() < 1e-3      # voxel centres: unit origin + offset vs grid origin + offset (float32)
    # what the dense volume updated OUTSIDE the allocated units is free space in front of the surface (tsdf clamped to 1): the band
    # around the zero crossing -- everything marching cubes needs -- lives in the allocated units
    out = (dense.weight > 0) & ~alloc
    assert (dense.tsdf[out] < 1.0).float().mean().item() < 0.02


def test_two_tile_volumes_merge_to_the_joint_volume():
    """extract_mesh_split.py:91-119 integrates the frames of every tile into ONE volume; here each 'tile' integrates its own frames on the
    device and the volumes are fused afterwards -- merge_from on one device, merge_() through torch.distributed (single-process group)."""
    import os
    import torch.distributed as dist
    frs = tsdf_cases.frames(6, seed=3)
    a, b = _hip_volume(frs[:3]), _hip_volume(frs[3:])
    a.merge_from(b)
    ref = _oracle_units(frs)
    co, t, w, c = (x.cpu().numpy() for x in a.units())
    want = {tuple(k): i for i, k in enumerate(ref[0].tolist())}
    assert set(map(tuple, co.tolist())) == set(want)
    nbad = ntot = 0
    for i, k in enumerate(co.tolist()):
        j = want[tuple(k)]
        nbad += int((w[i] != ref[2][j]).sum()); ntot += w[i].size
        same = w[i] == ref[2][j]
        nbad += int((np.abs(t[i][same] - ref[1][j][same]) > 2e-4).sum())
    assert nbad <= 2e-4 * ntot, (nbad, ntot)
    # the distributed entry point with a one-rank group (the N-rank path is covered by tests/test_dist_cpu.py on gloo and, when the
    # box has >= 2 GPUs, by tests/test_gpu_multi.py on RCCL)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        n0 = a.num_units
        a.merge_()
        assert a.num_units == n0
    finally:
        if own:
            dist.destroy_process_group()


def test_deferred_frames_equal_the_synchronous_ones_also_through_a_pool_growth():
    """integrate(defer=True) only enqueues (no host synchronisation); the outcome of a frame is handled by the next call that needs the volume.  Same units,
    bit-identical voxels -- also when a deferred frame runs out of pool slots (nothing of it is integrated; the next call grows the pool and runs it again)."""
    frs = tsdf_cases.frames(4)
    ref = _hip_volume(frs, cap=8192)
    for cap in (8192, 64):
        vol = _hip_volume(frs, cap=cap, defer=True)
        assert vol._pending is not None                       # the last frame is still in flight
        n = vol.num_units                                     # ... until somebody reads the volume
        assert vol._pending is None and n == ref.num_units and (cap == 8192 or vol.cap > 64)
        kr = {tuple(k): i for i, k in enumerate(ref.units()[0].tolist())}
        order = torch.tensor([kr[tuple(k)] for k in vol.units()[0].tolist()], device="cuda")
        for a, b in zip(vol.units()[1:], ref.units()[1:]):
            assert torch.equal(a, b[order])
        assert not torch.isnan(vol.units()[1]).any()


def _same_volume(vol, ref):
    kr = {tuple(k): i for i, k in enumerate(ref.units()[0].tolist())}
    assert vol.num_units == ref.num_units and set(kr) == set(map(tuple, vol.units()[0].tolist()))
    order = torch.tensor([kr[tuple(k)] for k in vol.units()[0].tolist()], device="cuda")
    for a, b in zip(vol.units()[1:], ref.units()[1:]):
        assert torch.equal(a, b[order])


def test_many_frames_in_flight_also_through_a_refused_frame():
    """Up to MAX_IN_FLIGHT deferred frames wait without the host looking at them; a frame that runs out of pool slots integrates nothing, neither do the
    frames enqueued behind it, and finish() grows the pool and runs them all again in order: bit-identical to the synchronous volume."""
    from gsrast.tsdf import ScalableTSDFVolume
    frs = tsdf_cases.frames(11, seed=5)
    ref = _hip_volume(frs, cap=16384)
    for cap in (16384, 128):
        vol = _hip_volume(frs, cap=cap, defer=True)
        assert 1 <= len(vol._queue) <= ScalableTSDFVolume.MAX_IN_FLIGHT
        _same_volume(vol, ref)
        assert not vol._queue and (cap == 16384 or vol.cap > 128)


def test_only_written_groups_exist_until_somebody_reads_the_pools():
    """ABI 8: a frame writes only the 128-byte lines it updates a group of (bits set in `mask`), also in a unit it opens; everything else keeps whatever the
    record held (NaN here) until units() / to_dense() materialise the volume.  The written groups, re-ordered from brick to x-major order, are the oracle's
    voxels.  Growth adds a chunk of records and copies no voxel: chunk 0 is the same memory before and after."""
    frs = tsdf_cases.frames(2)
    vol = _hip_volume(frs, cap=256)
    vol.finish()
    n = int(vol.counters[0].item())
    assert len(vol.chunks) > 1 and vol.chunks[0].shape[0] == 256 and sum(c.shape[0] for c in vol.chunks) == vol.cap >= n
    bits = ((vol.mask[:n].view(n, 16, 1) >> torch.arange(64, device="cuda").view(1, 1, 64)) & 1).bool().view(n, 1024)       # [unit, group]
    stamped = vol.stamp[:n] != 0
    w = vol.records(n)[:, 1].view(n, 1024, 4)
    first = stamped & (torch.arange(n, device="cuda") < 256)              # chunk 0 was poisoned before use (later chunks are fresh allocations)
    assert first.sum() > 100 and torch.isnan(w[first][~bits[first]]).all()      # never written: still the poison
    assert not torch.isnan(w[stamped][bits[stamped]]).any()
    assert 0.2 < bits[stamped].float().mean().item() < 0.95                  # a band through the units, not whole units
    runs = bits.view(n, 128, 8)
    assert (runs.all(-1) | ~runs.any(-1)).all()                            # whole 128-byte lines: eight groups written together (full-line write-back)
    assert (vol.records(n)[:, 0].view(n, 1024, 4)[stamped][bits[stamped]].abs() <= 1).all()
    _compare(vol, _oracle_units(frs))                                      # materialises
    assert not torch.isnan(vol.records(n)).any() and (vol.mask[:n] == -1).all()


def test_merge_reads_the_other_volumes_pools_in_place():
    """merge_from on one device (gsr_tsdf_sparse_merge_volume: storage order, written groups only) == merge_units_ with the exported plain arrays
    (gsr_tsdf_sparse_merge: what other ranks send), bit for bit, into an empty and into a populated volume."""
    from gsrast.tsdf import ScalableTSDFVolume
    frs = tsdf_cases.frames(6, seed=7)
    for base in (0, 2):
        a1 = _hip_volume(frs[:base], cap=8192); a2 = _hip_volume(frs[:base], cap=8192)
        b1 = _hip_volume(frs[base:], cap=8192); b2 = _hip_volume(frs[base:], cap=8192)
        a1.merge_from(b1)                                                  # pools in place (b1 never materialised)
        co, t, w, c = b2.units()
        a2.merge_units_(co, t, w, c, assume_unique=True)
        _same_volume(a1, a2)
    small = _poison(ScalableTSDFVolume(VL, TR, capacity_units=32))         # merging grows the pool like a frame does
    small.merge_from(b1)
    _same_volume(small, b2)
    fixed = ScalableTSDFVolume(VL, TR, capacity_units=32, auto_grow=False)
    with pytest.raises(RuntimeError, match="capacity exhausted"):
        fixed.merge_from(b1)
    assert fixed.num_units <= 32 and not fixed.units()[2].any()            # usable afterwards, no uninitialised pool memory on show
    with pytest.raises(RuntimeError, match="capacity exhausted"):          # ... and the failure flag does not stick
        fixed.merge_from(b1)


def test_capacity_overflow_raises_or_grows():
    """auto_grow=False: a frame that needs more units than the pool holds raises; the default doubles the pool until the frame fits and then
    holds exactly the volume a large-enough pool would have built."""
    from gsrast.tsdf import ScalableTSDFVolume
    f = tsdf_cases.frames(1)[0]
    args = (torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda(), f["fx"], f["fy"], f["cx"], f["cy"], f["E"])
    vol = _poison(ScalableTSDFVolume(VL, TR, capacity_units=16, auto_grow=False))
    with pytest.raises(RuntimeError, match="capacity exhausted"):
        vol.integrate(*args, depth_trunc=DT)
    assert vol.num_units <= 16 and not torch.isnan(vol.units()[2]).any() and not vol.units()[2].any()      # the units the failed frame allocated: explicit empty units, not pool garbage
    with pytest.raises(RuntimeError, match="capacity exhausted"):          # the volume stays usable: the next frame is refused for the same reason, not for a stale flag
        vol.integrate(*args, depth_trunc=DT)
    small = _poison(ScalableTSDFVolume(VL, TR, capacity_units=16))
    small.integrate(*args, depth_trunc=DT)
    big = ScalableTSDFVolume(VL, TR, capacity_units=8192)
    big.integrate(*args, depth_trunc=DT)
    assert small.cap > 16 and small.num_units == big.num_units
    key = lambda v: {tuple(k): i for i, k in enumerate(v.units()[0].tolist())}
    ks, kb = key(small), key(big)
    assert set(ks) == set(kb)
    order = torch.tensor([ks[k] for k in kb], device="cuda")
    for a, b in zip(small.units()[1:], big.units()[1:]):
        assert torch.equal(a[order], b)


def test_rejects_a_truncation_band_wider_than_four_units_and_samples_out_of_range():
    """a depth sample whose +-sdf_trunc box spans more than 4 units per axis, or whose unit coordinate leaves the 21-bit key range,
    used to be dropped silently (a volume with no units and no error)."""
    from gsrast.tsdf import ScalableTSDFVolume
    with pytest.raises(RuntimeError, match="1.5 units"):
        ScalableTSDFVolume(0.01, 0.3)                                      # 30 voxels > 24
    vol = ScalableTSDFVolume(1e-4, 5e-4, capacity_units=64)
    depth = torch.full((1, 8, 8), 2000.0, device="cuda"); rgb = torch.zeros(3, 8, 8, device="cuda")
    with pytest.raises(RuntimeError, match="outside the addressable volume"):          # 2000 / (16 * 1e-4) > 2^20 units from the origin
        vol.integrate(rgb, depth, 10.0, 10.0, 4.0, 4.0, np.eye(4, dtype=np.float32), depth_trunc=1e9)


def test_merge_units_fuses_duplicate_coordinates_and_colour_scales_agree():
    """merge_units_ with a caller-supplied list that names a unit twice (the merge kernel runs one workgroup per listed unit: duplicates would
    race) gives the weighted fusion; quantize_rgb8=False stores colours on the same 0..255 scale as the default."""
    from gsrast.tsdf import ScalableTSDFVolume
    g = torch.Generator().manual_seed(0)
    co = torch.tensor([[1, 2, 3], [0, 0, 0], [1, 2, 3]], dtype=torch.int32).cuda()
    t = torch.rand(3, 16, 16, 16, generator=g).cuda(); w = torch.randint(1, 4, (3, 16, 16, 16), generator=g).float().cuda()
    c = torch.rand(3, 16, 16, 16, 3, generator=g).cuda()
    vol = _poison(ScalableTSDFVolume(VL, TR, capacity_units=64))
    vol.merge_units_(co, t, w, c)
    uc, ut, uw, _ = vol.units()
    assert vol.num_units == 2
    k = [tuple(x) for x in uc.tolist()].index((1, 2, 3))
    assert torch.equal(uw[k], w[0] + w[2]) and torch.allclose(ut[k], (t[0] * w[0] + t[2] * w[2]) / (w[0] + w[2]), atol=1e-6)
    f = tsdf_cases.frames(1)[0]
    rgb = torch.from_numpy(f["rgb"]).cuda(); depth = torch.from_numpy(f["depth"]).cuda()
    a = ScalableTSDFVolume(VL, TR, capacity_units=8192); b = ScalableTSDFVolume(VL, TR, capacity_units=8192)
    a.integrate(rgb, depth, f["fx"], f["fy"], f["cx"], f["cy"], f["E"], depth_trunc=DT)
    b.integrate(rgb, depth, f["fx"], f["fy"], f["cx"], f["cy"], f["E"], depth_trunc=DT, quantize_rgb8=False)
    ka = {tuple(k): i for i, k in enumerate(a.units()[0].tolist())}
    order = torch.tensor([ka[tuple(k)] for k in b.units()[0].tolist()], device="cuda")          # the two pools number their units independently
    ca, cb = a.units()[3][order], b.units()[3]
    assert ca.max().item() > 1.5 and (ca - cb).abs().max().item() <= 1.0          # same scale, differing by the rounding to integers only


def test_reference_default_resolution_runs_without_a_dense_grid():
    """extract_mesh.py:125-128: voxel = depth_trunc / 1024, sdf_trunc = 5 voxels.  A dense grid of that resolution is >= 1024^3 voxels
    (21 GB); the sparse volume allocates only the band around the surface."""
    from gsrast.tsdf import ScalableTSDFVolume
    W, H = 480, 360
    depth_trunc = 8.0
    vl = depth_trunc / 1024
    vol = ScalableTSDFVolume(vl, 5 * vl, capacity_units=60000)
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    for k in range(3):
        depth = (4.0 + 0.4 * np.sin(u / 40.0 + k) + 0.3 * np.cos(v / 30.0)).astype(np.float32)[None]
        rgb = np.random.default_rng(k).uniform(0, 1, (3, H, W)).astype(np.float32)
        E = np.eye(4, dtype=np.float32); E[0, 3] = 0.02 * k
        vol.integrate(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), 420.0, 420.0, W / 2, H / 2, E, depth_trunc=depth_trunc)
    n = vol.num_units
    co, t, w, c = vol.units()
    assert 1000 < n < 60000
    assert (w > 0).any(dim=-1).any(dim=-1).any(dim=-1).float().mean().item() > 0.9          # nearly every opened unit received samples
    # surface voxels carry |tsdf| < 1 with both signs (a zero crossing exists for marching cubes)
    tt = t[w > 0]
    assert (tt < 0).any() and (tt > 0).any() and tt.abs().max().item() <= 1.0
