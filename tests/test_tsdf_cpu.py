"""CPU tests of the block-sparse TSDF oracle (oracle/gsr_oracle.c ref_tsdf_sparse_*) and of the device-agnostic merge helpers.
The oracle restates Open3D 0.18's ScalableTSDFVolume algorithm -- parity unpinned (Open3D is absent) -- so what is pinned here is
internal consistency: the sparse volume equals the dense uniform-volume restatement on every allocated unit, allocation follows the
truncation band of the sampled depth pixels, and fusing per-tile volumes equals integrating all frames into one."""
import numpy as np
import torch

import oracle
import tsdf_cases

VL, TR, DT = 0.02, 0.1, 6.0


def _sparse(frs):
    v = oracle.SparseTSDF(VL, TR)
    for f in frs:
        v.integrate(tsdf_cases.rgb8(f["rgb"]), f["depth"], f["fx"], f["fy"], f["cx"], f["cy"], f["E"], depth_trunc=DT)
    return v.units()


def test_sparse_equals_dense_on_allocated_units():
    frs = tsdf_cases.frames(3)
    co, t, w, c = _sparse(frs)
    assert 50 < len(co) < 4000 and len(np.unique(co, axis=0)) == len(co)
    lo, hi = co.min(0), co.max(0) + 1
    dims = tuple(int(d) * 16 for d in (hi - lo))
    T = np.zeros(dims, np.float32); Wt = np.zeros(dims, np.float32); C = np.zeros(dims + (3,), np.float32)
    origin = (lo * np.float32(VL * 16)).astype(np.float32)
    for f in frs:
        oracle.tsdf_integrate_dense(dims, origin, VL, TR, DT, f["depth"], tsdf_cases.rgb8(f["rgb"]), f["fx"], f["fy"], f["cx"], f["cy"], f["E"], T, Wt, C)
    touched_any = nbad = ntot = 0
    for k in range(len(co)):
        x, y, z = (co[k] - lo) * 16
        sl = (slice(x, x + 16), slice(y, y + 16), slice(z, z + 16))
        # a unit opened only from frame k on has missed the earlier frames in the sparse volume: weights never exceed the dense ones
        nbad += int((w[k] > Wt[sl]).sum())      # only where float32 voxel-centre rounding flips the sdf > -trunc gate
        same = w[k] == Wt[sl]
        # voxel centres: unit origin + offset here, grid origin + offset there (float32): a projection can round into the next pixel
        nbad += int((np.abs(t[k][same] - T[sl][same]) > 5e-5).sum() + (np.abs(c[k][same] - C[sl][same]).max(-1) > 0.05).sum()); ntot += int(same.sum())
        touched_any += int((w[k] > 0).any())
    assert touched_any > 0.5 * len(co) and nbad <= 2e-3 * ntot, (nbad, ntot)
    # every voxel the dense volume updated within the truncation band of a SAMPLED pixel lies in an allocated unit: the band is covered
    assert (Wt > 0).sum() > 0


def test_allocation_is_the_truncation_band_of_sampled_pixels():
    f = tsdf_cases.frames(1, holes=False)[0]
    co, t, w, c = _sparse([f])
    E = f["E"].astype(np.float64); P = np.linalg.inv(E)
    H, W = f["depth"].shape[-2:]
    want = set()
    unit = np.float32(VL * 16)
    for v in range(0, H, 4):
        for u in range(0, W, 4):
            d = float(f["depth"][0, v, u])
            pc = np.array([(u - f["cx"]) * d / f["fx"], (v - f["cy"]) * d / f["fy"], d, 1.0])
            p = (P @ pc)[:3]
            lo = np.floor((p - TR) / unit).astype(int); hi = np.floor((p + TR) / unit).astype(int)
            for x in range(lo[0], hi[0] + 1):
                for y in range(lo[1], hi[1] + 1):
                    for z in range(lo[2], hi[2] + 1):
                        want.add((x, y, z))
    got = set(map(tuple, co.tolist()))
    # float32 vs float64 rounding can move a point across a unit face: allow a sliver
    assert len(got ^ want) <= max(2, len(want) // 200), (len(got), len(want), len(got ^ want))


def test_merge_unit_lists_equals_joint_integration():
    from gsrast.tsdf import merge_unit_lists
    frs = tsdf_cases.frames(4)
    coA, tA, wA, cA = _sparse(frs[:2])
    coB, tB, wB, cB = _sparse(frs[2:])
    co, t, w, c = _sparse(frs)
    f = lambda a: torch.from_numpy(a.reshape(a.shape[0], 4096, *a.shape[4:]))
    mco, mt, mw, mc = merge_unit_lists(torch.cat([torch.from_numpy(coA), torch.from_numpy(coB)]), torch.cat([f(tA), f(tB)]),
                                       torch.cat([f(wA), f(wB)]), torch.cat([f(cA), f(cB)]))
    ref = {tuple(k): i for i, k in enumerate(co.tolist())}
    assert set(map(tuple, mco.tolist())) == set(ref)
    for i, k in enumerate(mco.tolist()):
        j = ref[tuple(k)]
        assert np.array_equal(mw[i].numpy(), w[j].reshape(-1))
        assert np.abs(mt[i].numpy() - t[j].reshape(-1)).max() < 1e-5
        assert np.abs(mc[i].numpy() - c[j].reshape(-1, 3)).max() < 1e-2


def test_brick_storage_order_is_the_formula_of_the_header():
    """include/gsrast.h (ABI 8): voxel (x, y, z) of a unit lives at float index 4 g + (z & 3) of its plane,
    g = (x>>2)<<8 | (y>>2)<<6 | (z>>2)<<4 | ((x>>1)&1)<<3 | ((y>>1)&1)<<2 | (x&1)<<1 | (y&1).  gsrast.tsdf.brick_to_xmajor -- what units() hands out -- inverts exactly
    that; a 128-byte line (8 groups = 32 floats) is a 2 x 4 x 4 box of voxels and a 64-byte sector a 2 x 2 x 4 one (what the layout is for)."""
    from gsrast.tsdf import brick_to_xmajor
    x, y, z = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij")
    g = ((x >> 2) << 8) | ((y >> 2) << 6) | ((z >> 2) << 4) | (((x >> 1) & 1) << 3) | (((y >> 1) & 1) << 2) | ((x & 1) << 1) | (y & 1)
    idx = 4 * g + (z & 3)
    assert sorted(idx.ravel().tolist()) == list(range(4096))
    plane = torch.arange(4096, dtype=torch.float32).reshape(1, 4096)          # the value of a float = its storage index
    out = brick_to_xmajor(plane)[0].numpy().astype(np.int64)
    assert np.array_equal(out, idx)
    extent = lambda sel: tuple(int(a[sel].max() - a[sel].min() + 1) for a in (x, y, z))
    for line in (0, 1, 77, 127):
        assert extent(idx // 32 == line) == (2, 4, 4)
    for sector in (0, 3, 200, 255):
        assert extent(idx // 16 == sector) == (2, 2, 4)
    planes = torch.arange(2 * 5 * 4096, dtype=torch.float32).reshape(2, 5, 4096)        # leading dimensions pass through
    assert brick_to_xmajor(planes).shape == (2, 5, 16, 16, 16) and torch.equal(brick_to_xmajor(planes)[1, 3], brick_to_xmajor(planes[1, 3]))
