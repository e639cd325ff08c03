"""Parity criterion against a float64 truth.  TEST INFRASTRUCTURE ONLY; pure numpy, so the checker itself is tested on CPU.

Inputs, all for ONE case (same scene, same upstream gradients):
  cand   the implementation under test (the HIP library; in the CPU self-tests an oracle build stands in for it)
  f32    oracle/libgsr_oracle.so      -- the reference's formulas in float32, every operation rounded on its own
  fma    oracle/libgsr_oracle_fma.so  -- the same with FMA contraction (what nvcc does for the real reference); may be None
  truth  oracle/libgsr_oracle_f64.so  -- the same statements in float64 on the integer stages of `f32`, plus the gate bookkeeping:
         margin / gate / splat per pixel (oracle.Truth)

What is asserted.  The blend has discrete gates (alpha < 1/255, T(1-alpha) < 1e-4, T > 0.5, rho3d <= rho2d, depth < near, power > 0).  A pixel is
ROBUST when every gate decision the truth took lies further from its threshold than the first-order float32 error bound of the tested quantity
(margin > 1): any correct float32 evaluation takes the same decisions there.  On robust pixels
  * integer outputs (last contributor, median contributor, median splat id) equal the truth's EXACTLY, no exception;
  * float maps: |cand - truth| <= tol (1e-4, scaled by max(1, max|truth|)) except on at most 2 x the pixels where the float32 ORACLE itself misses that bar
    against the truth (+2: edge-on surfels amplify one-ulp differences without any gate), and the L2 / max error of cand is at most 2 x / 4 x the oracle's.
On FRAGILE pixels (margin <= 1; every one carries the name of its closest gate) a flip is legitimate: only their number is bounded
(<= 2 x the float32 oracle's own flips + a few), and it is reported per gate.
Gradients, per tensor, rows of Gaussians that are not the flipping splat of a fragile pixel ("robust rows"):
  * relative L2 error vs the truth <= max(1e-3, 2 x the float32 oracle's relative L2 error vs the truth);
  * fraction of elements beyond 1e-3 |truth| + 1e-3 rms(truth) <= 2 x the oracle's fraction + 1e-4 (at least two elements);
and on the remaining rows relative L2 <= 4 x the oracle's + 1e-3.  A 5 % error in any gradient tensor, a 3e-4 offset in any map or a shifted
contributor index trips these bars (tests/test_truth_cpu.py).

THE FLOOR (round 5).  truth["floor"] is a second float64 run that blends -- forward and backward, in float64 -- the per-gaussian state of the FLOAT32
run (transMat / conic, normal, opacity, projected centre, depth, SH colour exactly as a float32 preprocess leaves them in the reference's geomBuffer;
oracle.Truth(f32_geometry=True)).  Its distance from the truth is the error the reference's float32 preprocess imposes on EVERY implementation that
blends float32 per-gaussian state, the reference's own CUDA kernels included ("floor_rel_l2"); the candidate's distance from the floor run
("rel_l2_vs_floor") is the error of its blend arithmetic proper.  With a floor at hand the bars on the robust set are the NOMINAL ones of north_star,
with no relative term, against the floor run:
  * maps: at most 2 (+1e-6 N) robust pixels beyond 1e-4 vs the floor run;
  * gradients: relative L2 vs the floor run <= 1e-3, and relative L2 vs the truth <= max(1e-3, 1.25 x floor_rel_l2).
Measured at BASELINE size (profiles/r05_full_size_parity.jsonl, 14 cases): HIP vs the floor run 5e-7 ... 4.4e-5 on every gradient tensor of every variant, 0 robust
pixels beyond 1e-4 on every map; the surfel's geometric gradients sit 1.4e-3 ... 1.7e-2 from the truth because the FLOOR does (equal to three digits)."""
import numpy as np

TOL_IMG, TOL_GRAD = 1e-4, 1e-3


def _rel(a, b):
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def check_map(name, cand, oracles, truth, robust, tol=TOL_IMG, report=None, floor=None):
    """One float map (any leading shape, last two dims H, W) against the truth.  floor: the same map of the float32-geometry float64 run (reported)."""
    t = np.asarray(truth, np.float64)
    c = np.abs(np.asarray(cand, np.float64) - t)
    os_ = [np.abs(np.asarray(o, np.float64) - t) for o in oracles if o is not None]
    s = tol * max(1.0, float(np.abs(t).max()))
    rb = np.broadcast_to(robust, t.shape)
    n = t.size
    bad_c = c > s
    bad_o = [o > s for o in os_]
    nb_c_r = int((bad_c & rb).sum()); nb_o_r = max(int((b & rb).sum()) for b in bad_o)
    nb_c_f = int((bad_c & ~rb).sum()); nb_o_f = max(int((b & ~rb).sum()) for b in bad_o)
    l2_c = float(np.sqrt((c[rb] ** 2).sum())); l2_o = max(float(np.sqrt((o[rb] ** 2).sum())) for o in os_)
    mx_c = float(c[rb].max()) if rb.any() else 0.0
    mx_o = max(float(o[rb].max()) if rb.any() else 0.0 for o in os_)
    rep = dict(tol=s, robust_px_beyond_tol=nb_c_r, oracle_robust_px_beyond_tol=nb_o_r, fragile_px_beyond_tol=nb_c_f,
               oracle_fragile_px_beyond_tol=nb_o_f, robust_l2=l2_c, oracle_robust_l2=l2_o, robust_max=mx_c, oracle_robust_max=mx_o)
    if floor is not None:
        fl = np.asarray(floor, np.float64)
        ef = np.abs(fl - t); ec = np.abs(np.asarray(cand, np.float64) - fl)
        rep.update(floor_robust_px_beyond_tol=int(((ef > s) & rb).sum()), floor_robust_max=float(ef[rb].max()) if rb.any() else 0.0,
                   robust_px_beyond_tol_vs_floor=int(((ec > s) & rb).sum()), robust_max_vs_floor=float(ec[rb].max()) if rb.any() else 0.0)
    if report is not None:
        report[name] = rep
    if floor is not None:      # the nominal bar, no relative term: against the exact blend of the same float32 per-gaussian state every robust pixel is within tol
        assert rep["robust_px_beyond_tol_vs_floor"] <= 2 + int(1e-6 * n), (f"{name}: {rep['robust_px_beyond_tol_vs_floor']} ROBUST pixels beyond {s:.1e} vs the float64 "
                                                                          f"blend of the float32 geometry (max {rep['robust_max_vs_floor']:.2e})")
    assert nb_c_r <= 2 * nb_o_r + 2 + int(1e-6 * n), f"{name}: {nb_c_r} ROBUST pixels beyond {s:.1e} vs the float64 truth (float32 oracle: {nb_o_r})"
    assert l2_c <= 2.0 * l2_o + 1e-7 * np.sqrt(n), f"{name}: L2 error on robust pixels {l2_c:.3e} > 2 x the float32 oracle's {l2_o:.3e}"
    assert mx_c <= 4.0 * mx_o + s, f"{name}: max error on robust pixels {mx_c:.3e} (float32 oracle {mx_o:.3e})"
    assert nb_c_f <= 2 * nb_o_f + 8 + int(1e-5 * n), f"{name}: {nb_c_f} fragile pixels beyond {s:.1e} (float32 oracle: {nb_o_f})"
    return rep


def check_index(name, cand, f32, truth, robust, report=None):
    """An integer-valued per-pixel output: exact on robust pixels, flips bounded on fragile ones."""
    c = np.asarray(cand) != np.asarray(truth)
    o = np.asarray(f32) != np.asarray(truth)
    rb = np.broadcast_to(robust, c.shape)
    rep = dict(robust_mismatches=int((c & rb).sum()), fragile_mismatches=int((c & ~rb).sum()), oracle_fragile_mismatches=int((o & ~rb).sum()))
    if report is not None:
        report[name] = rep
    assert rep["robust_mismatches"] == 0, f"{name}: {rep['robust_mismatches']} mismatches on ROBUST pixels (no float32 gate within its error bound): logic divergence"
    assert rep["fragile_mismatches"] <= 2 * rep["oracle_fragile_mismatches"] + 8 + int(1e-5 * c.size), (name, rep)
    return rep


def robust_rows(truth_splat, fragile, P):
    """Gaussians that are not the flipping splat of any fragile pixel."""
    marg = np.zeros(P, bool)
    ids = np.asarray(truth_splat)[fragile]
    marg[ids[ids >= 0]] = True
    return ~marg


FLOOR_SLACK = 1.25


def check_grad(name, cand, oracles, truth, rows, tol=TOL_GRAD, report=None, floor=None):
    t = np.asarray(truth, np.float64).reshape(truth.shape[0], -1)
    c = np.asarray(cand, np.float64).reshape(t.shape)
    os_ = [np.asarray(o, np.float64).reshape(t.shape) for o in oracles if o is not None]
    tr, cr = t[rows], c[rows]
    rms = float(np.sqrt((tr * tr).mean())) if tr.size else 0.0
    tol_el = tol * np.abs(tr) + tol * rms
    l2_c = _rel(cr, tr); l2_o = max(_rel(o[rows], tr) for o in os_)
    fr_c = float((np.abs(cr - tr) > tol_el).mean()) if tr.size else 0.0
    fr_o = max(float((np.abs(o[rows] - tr) > tol_el).mean()) if tr.size else 0.0 for o in os_)
    rest = ~rows
    l2_cn = _rel(c[rest], t[rest]) if rest.any() else 0.0
    l2_on = max(_rel(o[rest], t[rest]) if rest.any() else 0.0 for o in os_)
    rep = dict(rel_l2=l2_c, oracle_rel_l2=l2_o, frac_elements_beyond=fr_c, oracle_frac_elements_beyond=fr_o, rel_l2_flip_rows=l2_cn,
               oracle_rel_l2_flip_rows=l2_on, rel_l2_all_rows=_rel(c, t), oracle_rel_l2_all_rows=max(_rel(o, t) for o in os_))
    if floor is not None:
        fl = np.asarray(floor, np.float64).reshape(t.shape)
        rep.update(floor_rel_l2=_rel(fl[rows], tr), rel_l2_vs_floor=_rel(cr, fl[rows]), oracle_rel_l2_vs_floor=max(_rel(o[rows], fl[rows]) for o in os_))
    if report is not None:
        report[name] = rep
    if floor is not None:
        assert l2_c <= max(tol, FLOOR_SLACK * rep["floor_rel_l2"]), (f"{name}: relative L2 vs the float64 truth {l2_c:.3e} (bar max({tol:.0e}, {FLOOR_SLACK} x the "
                                                                    f"float32-geometry floor {rep['floor_rel_l2']:.3e}))")
        assert rep["rel_l2_vs_floor"] <= tol, f"{name}: relative L2 vs the float64 blend of the float32 geometry {rep['rel_l2_vs_floor']:.3e} > {tol:.0e}"
    assert l2_c <= max(tol, 2.0 * l2_o), f"{name}: relative L2 vs the float64 truth {l2_c:.3e} (bar max({tol:.0e}, 2 x float32 oracle {l2_o:.3e}))"
    assert fr_c <= 2.0 * fr_o + max(1e-4, 2.0 / max(tr.size, 1)), f"{name}: {fr_c:.2e} of the elements beyond {tol}|t| + {tol} rms (float32 oracle {fr_o:.2e})"
    assert l2_cn <= 4.0 * l2_on + tol, f"{name}: rows of flipping splats: relative L2 {l2_cn:.3e} (float32 oracle {l2_on:.3e})"
    return rep


GRAD_PAIRS = [("dL_dmeans3D", "dL_dmeans3D"), ("dL_dscales", "dL_dscales"), ("dL_drotations", "dL_drotations"), ("dL_dopacities", "dL_dopacity"),
              ("dL_dmeans2D", "dL_dmeans2D")]


def check_case(variant, cm, cand, f32, fma, truth, report=None, min_robust=0.9):
    """cand / f32 / fma: dicts with color, final_T [k,H,W], n_contrib [k2,H,W], others | all_map, plane_depth, observe, grads (oracle naming for
    f32 / fma / truth, product naming -- dL_dopacities, dL_dshs, dL_dcolors_precomp -- for cand).  truth additionally: margin, gate, splat."""
    report = {} if report is None else report
    robust = truth["margin"] > 1.0
    N = robust.size
    report["robust_pixel_fraction"] = float(robust.mean())
    from oracle import GATE_NAMES
    report["fragile_pixels_by_gate"] = {GATE_NAMES[int(k)]: int(((truth["gate"] == k) & ~robust).sum()) for k in np.unique(truth["gate"][~robust])}
    # (min_robust: 0.9 in the suite; the P = 1 000 000 report case has 3.3 x the gate decisions per pixel and 0.88 of its pixels robust -- tools/full_parity_report.py states its bar)
    assert robust.mean() >= min_robust, "the robust set must cover the image (the criterion would be vacuous)"
    orc = lambda key, idx=None: [None if o is None else (o[key] if idx is None else o[key][idx]) for o in (f32, fma)]
    flo = truth.get("floor")
    fl = lambda key, idx=None: None if flo is None else (flo[key] if idx is None else flo[key][idx])
    check_index("n_contrib", cand["n_contrib"][0], f32["n_contrib"][0], truth["n_contrib"][0], robust, report)
    check_map("color", cand["color"], orc("color"), truth["color"], robust, report=report, floor=fl("color"))
    check_map("final_T", cand["final_T"][0], orc("final_T", 0), truth["final_T"][0], robust, report=report, floor=fl("final_T", 0))
    if variant == "surfel":
        check_index("median_contributor", cand["n_contrib"][1], f32["n_contrib"][1], truth["n_contrib"][1], robust, report)
        check_index("median_splat", cand["others"][7], f32["others"][7], truth["others"][7], robust, report)
        for ch in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10):
            check_map(f"others[{ch}]", cand["others"][ch], orc("others", ch), truth["others"][ch], robust, report=report, floor=fl("others", ch))
        check_map("M1", cand["final_T"][1], orc("final_T", 1), truth["final_T"][1], robust, report=report, floor=fl("final_T", 1))
        check_map("M2", cand["final_T"][2], orc("final_T", 2), truth["final_T"][2], robust, report=report, floor=fl("final_T", 2))
    if variant == "plane":
        check_map("all_map", cand["all_map"], orc("all_map"), truth["all_map"], robust, report=report, floor=fl("all_map"))
        # plane depth = dist / -(n . ray + 1e-8): unbounded where the rendered normal is orthogonal to the ray; compared where it is conditioned
        t = np.asarray(truth["all_map"], np.float64)
        H, W = robust.shape
        ok = robust & (np.abs(truth["plane_depth"][0]) < 1e3)
        check_map("plane_depth", cand["plane_depth"][0], orc("plane_depth", 0), truth["plane_depth"][0], ok, report=report, floor=fl("plane_depth", 0))
        d = np.abs(np.asarray(cand["observe"], np.int64) - np.asarray(truth["observe"], np.int64))
        nf = int((~robust).sum())
        report["observe"] = dict(sum_abs_diff=int(d.sum()), splats_differing=int((d > 0).sum()), fragile_pixels=nf)
        assert d.sum() <= nf and d.max() <= 4, f"out_observe differs by {int(d.sum())} in total with {nf} fragile pixels"
    rows = robust_rows(truth["splat"], ~robust, truth["grads"]["dL_dmeans3D"].shape[0])
    report["robust_row_fraction"] = float(rows.mean())
    pairs = list(GRAD_PAIRS) + [("dL_dshs", "dL_dsh") if cm == "sh" else ("dL_dcolors_precomp", "dL_dcolors")]
    if variant == "plane":
        pairs += [("dL_dall_map", "dL_dall_map"), ("dL_dmeans2D_abs", "dL_dmeans2D_abs")]
    for a, b in pairs:
        ca = cand["grads"][a if a in cand["grads"] else b]
        check_grad(b, ca, [f32["grads"][b], None if fma is None else fma["grads"][b]], truth["grads"][b], rows, report=report,
                   floor=None if flo is None else flo["grads"][b])
    return report


def oracle_outputs(f, grads):
    """dict of an oracle.Forward / oracle.Truth in the layout check_case expects."""
    ft, nc = f.image_state()
    d = dict(color=f.color.copy(), final_T=ft, n_contrib=nc, grads=grads)
    if f.others is not None:
        d["others"] = f.others.copy()
    if f.out_all_map is not None:
        d.update(all_map=f.out_all_map.copy(), plane_depth=f.plane_depth.copy(), observe=f.observe.copy())
    for k in ("margin", "gate", "splat", "splat_noise"):
        if hasattr(f, k):
            d[k] = getattr(f, k)
    return d


def run_oracles(sc, variant, og, hip_state=None, floor=True, fma=True):
    """-> (f32, fma, truth) output dicts + the float32 integer stages (for the bit-exact checks of the caller).  hip_state (hiprun.run_raw): the HIP
    library's filtered tile-instance list is held against the oracle's while both oracle runs are alive (tests/tile_cull.py reference_view: subset in
    order, nothing contributing dropped -- float32 and float64 --, exact float64 region kept); ints["view"] then carries the HIP n_contrib mapped to
    positions in the oracle's list."""
    import oracle
    if fma:
        with oracle.fma_twin():
            with oracle.Forward(sc, variant) as f2:
                fma = oracle_outputs(f2, f2.backward(**og))
    else:
        fma = None      # (BASELINE-size cases in the GPU suite: the float64 floor run is the yardstick there, the second float32 build only widens the relative bars)
    with oracle.Forward(sc, variant) as f:
        f32 = oracle_outputs(f, f.backward(**og))
        ints = dict(R=f.R, radii=f.radii.copy(), tiles_touched=f.tiles_touched(), point_list=f.point_list(), ranges=f.ranges())
        with oracle.Truth(sc, variant, f) as t:
            truth = oracle_outputs(t, t.backward(**og))
            if hip_state is not None:
                import tile_cull
                ints["view"] = tile_cull.reference_view(hip_state, f, truth=t, variant=variant)
        if floor:
            with oracle.Truth(sc, variant, f, f32_geometry=True) as t2:
                truth["floor"] = oracle_outputs(t2, t2.backward(**og))
    return f32, fma, truth, ints
