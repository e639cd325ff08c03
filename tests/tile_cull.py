"""Test-side view of the HIP library's tile-instance list (TEST INFRASTRUCTURE).

The reference emits one instance per tile of a gaussian's bounding rect (3DGS rasterizer_impl.cu:70-111, auxiliary.h:46-56).  The HIP library
drops, at emission, the instances whose gaussian cannot reach alpha >= 1/255 on any pixel of the tile (gs-sr_amd/csrc/gsr_tile_cull.h;
GSR_TILE_CULL=0 keeps them all).  Every output is unchanged by that; the PRIVATE indices that are positions in a tile's list (point_list,
n_contrib) are positions in the filtered list.  This module holds the filtered list against the oracle's full list:

  reference_view(st, f)   asserts  (1) the HIP list is the oracle's list restricted to a subset `keep`, order preserved, tile by tile;
                                   (2) no dropped instance passes the reference's alpha gate on any pixel of its tile (oracle
                                       ref_instance_max_alpha: the float64 build when `truth` is given, else the float32 build);
                                   (3) tiles_touched == per-gaussian count of kept instances, ranges == prefix of the per-tile counts;
                          returns  the mask and n_contrib mapped back to positions in the oracle's list, so that the index parity
                                   checks compare like with like.
  region_filter64(...)    an independent float64 numpy evaluation of "can the alpha >= 1/255 region meet the tile's pixel rectangle", with no
                          safety inflation: the exact region filter F.  Contributing instances C, F and the kept set K must nest: C <= F <= K.
"""
import numpy as np


def _tiles_of(ranges, R):
    ranges = np.asarray(ranges).astype(np.int64)
    lens = np.maximum(ranges[:, 1] - ranges[:, 0], 0)
    t = np.repeat(np.arange(ranges.shape[0], dtype=np.int64), lens)
    assert t.shape[0] == R, (t.shape[0], R)
    return t


def reference_view(st, f, truth=None, P=None, variant=None):
    """st: hiprun.run_raw state (HIP), f: oracle.Forward of the same scene, truth: optional oracle.Truth; variant: also hold the kept set against
    the independent float64 region filter.  Returns dict(keep, n_contrib [k2,H,W] in the reference's list positions, statistics)."""
    P = int(f.P if P is None else P)
    ref_list = f.point_list().astype(np.int64); ref_ranges = f.ranges().astype(np.int64)
    ref_tile = _tiles_of(ref_ranges, f.R)
    hip_list = st["point_list"].astype(np.int64); hip_tile = st["tile_keys"].astype(np.int64)
    R = int(st["R"])
    assert hip_list.shape[0] == R and R <= f.R, (R, f.R)
    # ---- (1) subsequence, order preserved: (tile, id) is unique in either list.  On the bucket-sort path hip_tile is rebuilt from the library's `ranges`
    # (k_debug_tile_keys): the check is not circular -- a boundary of `ranges` that is off by one entry hands that entry's id to the neighbouring tile, and
    # the pair (neighbour, id) is then either not in the reference's list, or a second copy of a pair the list already holds, or sits at the wrong
    # position of the neighbour's depth order: the three assertions below.
    pairs = hip_tile * P + hip_list
    assert np.unique(pairs).shape[0] == R, "a (tile, gaussian) pair appears twice in the HIP list"
    keep = np.isin(ref_tile * P + ref_list, pairs)
    assert int(keep.sum()) == R, f"{R} instances emitted, {int(keep.sum())} of them found in the reference's list"
    assert np.array_equal(ref_list[keep], hip_list), "HIP instance list is not the reference's list restricted to a subset (order or members differ)"
    assert np.array_equal(ref_tile[keep], hip_tile)
    # ---- (3) integer bookkeeping of the filtered list
    assert np.array_equal(st["tiles_touched"].astype(np.int64), np.bincount(hip_list, minlength=P)), "tiles_touched != kept instances per gaussian"
    T = ref_ranges.shape[0]
    counts = np.bincount(hip_tile, minlength=T)
    hr = st["ranges"].astype(np.int64)
    assert np.array_equal(np.maximum(hr[:, 1] - hr[:, 0], 0), counts), "ranges != histogram of the tile keys"
    nz = counts > 0
    assert np.array_equal(hr[nz, 0], (np.cumsum(counts) - counts)[nz])
    # ---- (2) nothing that contributes was dropped
    dropped = ~keep
    if dropped.any():
        if truth is not None:          # float64 evaluation of every (pixel, instance) pair of the dropped set's tiles
            a64 = truth.instance_max_alpha()
            bad = dropped & (a64 >= 1.0 / 255.0)
            assert not bad.any(), f"{int(bad.sum())} dropped instances pass the alpha gate in the float64 truth, max alpha {a64[bad].max():.4g}"
        else:                          # the float32 oracle's own evaluation
            amax = f.instance_max_alpha()
            bad = dropped & (amax >= 1.0 / 255.0)
            assert not bad.any(), f"{int(bad.sum())} dropped instances pass the reference's alpha gate (float32 oracle), max alpha {amax[bad].max():.4g}"
    stats = dict(R_reference=int(f.R), R_emitted=R, dropped_frac=float(dropped.mean()) if f.R else 0.0)
    if variant is not None and f.R:
        F = region_filter64(f, variant)
        missing = F & dropped
        # K >= F: the library's region carries safety inflation on top of the exact one, so an instance the exact region reaches must have been kept
        assert not missing.any(), f"{int(missing.sum())} instances inside the exact float64 region were dropped"
        stats.update(region_frac=float(F.mean()), kept_outside_region_frac=float((keep & ~F).mean()))
    # ---- n_contrib (1-based position in the tile's filtered list, 0 = none) -> position in the reference's list
    kept_idx = np.flatnonzero(keep)                                   # HIP global index -> reference global index
    H, W = st["n_contrib"].shape[-2:]
    gx = (W + 15) // 16
    tile_of_px = (np.arange(H)[:, None] // 16) * gx + np.arange(W)[None] // 16
    nc = st["n_contrib"].astype(np.int64)
    out = np.zeros_like(nc)
    for k in range(nc.shape[0]):
        c = nc[k]
        has = c > 0
        gi = hr[tile_of_px, 0] + c - 1
        assert (gi[has] < R).all() and (c[has] <= counts[tile_of_px][has]).all(), "n_contrib points past the tile's list"
        ri = kept_idx[np.where(has, gi, 0)]
        out[k] = np.where(has, ri - ref_ranges[tile_of_px, 0] + 1, 0)
    return dict(keep=keep, n_contrib=out.astype(np.uint32), **stats)


# ---------------------------------------------------------------------------------------------------- independent float64 region filter
def _rect_min_conic(A, B, C, X0, Y0, ext):
    """min over [X0, X0+ext] x [Y0, Y0+ext] of A x^2 + 2 B x y + C y^2 (positive-definite form), vectorised float64."""
    X1, Y1 = X0 + ext, Y0 + ext
    inside = (X0 <= 0) & (X1 >= 0) & (Y0 <= 0) & (Y1 >= 0)
    q = np.full(A.shape, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        for X in (X0, X1):
            y = np.clip(-B * X / C, Y0, Y1)
            q = np.minimum(q, A * X * X + 2 * B * X * y + C * y * y)
        for Y in (Y0, Y1):
            x = np.clip(-B * Y / A, X0, X1)
            q = np.minimum(q, A * x * x + 2 * B * x * Y + C * Y * Y)
    return np.where(inside, 0.0, q)


def region_filter64(f, variant):
    """-> bool[R] over the oracle's list: True where the region {alpha >= 1/255 possible} of the instance's gaussian, evaluated in float64 from the
    float32 oracle's per-gaussian state with NO safety inflation, meets the continuous rectangle of the tile's pixel centres.  EWA / PLANE: the conic
    q(d) <= 2 ln(255 o) (3DGS forward.cu:336-346).  SURFEL: the low-pass disc 2 |d|^2 <= 2 ln(255 o) (SURFEL forward.cu:369-372) or the projected
    contour rho3d <= 2 ln(255 o) of the splat, an ellipse from the dual conic sum_i t_i T_i T_i^T, t = (tt, tt, -1) -- the quantities compute_aabb
    (forward.cu:119-145) takes its centre and extents from; where that projection is not a proper ellipse in front of the camera the instance is
    kept (unknown)."""
    g = f.geom()
    ref_list = f.point_list().astype(np.int64)
    ref_tile = _tiles_of(f.ranges(), f.R)
    gx = (f.W + 15) // 16
    ox = (ref_tile % gx).astype(np.float64) * 16.0; oy = (ref_tile // gx).astype(np.float64) * 16.0
    co = g["conic_opacity"].astype(np.float64)[ref_list]
    m2 = g["means2D"].astype(np.float64)[ref_list]
    opa = co[:, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        tt = 2.0 * np.log(255.0 * opa)
    can = 255.0 * opa > 1.0
    if variant in ("ewa", "plane", 0, 2):
        q = _rect_min_conic(co[:, 0], co[:, 1], co[:, 2], ox - m2[:, 0], oy - m2[:, 1], 15.0)
        pd = (co[:, 0] > 0) & (co[:, 2] > 0) & (co[:, 0] * co[:, 2] - co[:, 1] ** 2 > 0)
        return can & (~pd | ~(q > tt))
    # SURFEL
    ex0, ey0 = ox - m2[:, 0], oy - m2[:, 1]
    ddx = np.maximum(np.maximum(ex0, -(ex0 + 15.0)), 0.0); ddy = np.maximum(np.maximum(ey0, -(ey0 + 15.0)), 0.0)
    disc = 2.0 * (ddx * ddx + ddy * ddy) <= tt
    Tm = g["cov"].astype(np.float64)[ref_list]
    Tu, Tv, Tw = Tm[:, 0:3], Tm[:, 3:6], Tm[:, 6:9]
    t3 = np.stack([tt, tt, -np.ones_like(tt)], axis=1)
    qd = lambda X, Y: (t3 * X * Y).sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = qd(Tw, Tw)
        cx, cy = qd(Tu, Tw) / w, qd(Tv, Tw) / w
        sxx = cx * cx - qd(Tu, Tu) / w; syy = cy * cy - qd(Tv, Tv) / w; sxy = cx * cy - qd(Tu, Tv) / w
        det = sxx * syy - sxy * sxy
        A, B, Cc = syy / det, -sxy / det, sxx / det
        q = _rect_min_conic(A, B, Cc, ox - cx, oy - cy, 15.0)
    proper = (w < 0) & (Tw[:, 2] > 0) & (sxx > 0) & (syy > 0) & (det > 0) & np.isfinite(q)
    return can & (disc | ~proper | ~(q > 1.0))
