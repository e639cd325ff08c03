"""CPU: the float64 build of the oracle (oracle/libgsr_oracle_f64.so, oracle.Truth) and the parity criterion built on it (tests/parity_truth.py).
The criterion is what `-m gpu` holds the HIP library to at BASELINE size; here an oracle build stands in for the candidate so that the checker
itself is tested: it must accept two correct float32 evaluations and reject a 5 % gradient error, a 3e-4 image offset and a shifted index."""
import copy

import numpy as np
import pytest

import oracle
import parity_truth as pt
import scenes


@pytest.fixture(scope="module", params=["ewa", "surfel", "plane"])
def case(request):
    variant = request.param
    W, H, P = 272, 176, 9000
    sc = scenes.make_scene(variant, P, W, H, seed=3, bg=(0.1, 0.3, 0.2))
    og = scenes.random_out_grads(variant, W, H, seed=3, scale=1.0)
    return (variant,) + pt.run_oracles(sc, variant, og)


def _as_cand(o):
    c = copy.deepcopy(o)
    return c


def test_truth_is_the_float64_value_of_the_float32_run(case):
    variant, f32, fma, truth, ints = case
    # same integer stages by construction; float outputs of the float32 build agree with the truth to float32 accuracy
    assert np.abs(f32["color"] - truth["color"]).max() < 5e-3 and np.median(np.abs(f32["color"] - truth["color"])) < 2e-7
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dcolors"):
        assert pt._rel(f32["grads"][k].astype(np.float64), truth["grads"][k]) < 1e-2
    # every pixel got a margin; the gate of a fragile pixel is named
    frag = truth["margin"] <= 1.0
    assert (truth["gate"][frag] > 0).all() and (truth["splat"][frag] >= 0).all()
    assert 0.0 < frag.mean() < 0.1


def test_disagreements_of_the_float32_builds_sit_on_fragile_pixels(case):
    """Where the float32 oracle takes a different discrete decision than the truth (last / median contributor), the truth had flagged the pixel."""
    variant, f32, fma, truth, ints = case
    frag = truth["margin"] <= 1.0
    bad = (f32["n_contrib"] != truth["n_contrib"]).any(axis=0)
    assert not (bad & ~frag).any()


def test_criterion_accepts_correct_float32_evaluations(case):
    variant, f32, fma, truth, ints = case
    rep = pt.check_case(variant, "precomp", _as_cand(fma), f32, None, truth)          # the FMA build as the candidate ...
    assert rep["robust_pixel_fraction"] > 0.9 and rep["robust_row_fraction"] > 0.8
    pt.check_case(variant, "precomp", _as_cand(f32), f32, fma, truth)                  # ... and the plain build against both


@pytest.mark.parametrize("what", ["grad5pct", "grad_one_component", "image_offset", "index_shift", "opacity_grad_scale"])
def test_criterion_rejects_defects(case, what):
    variant, f32, fma, truth, ints = case
    c = _as_cand(f32)
    if what == "grad5pct":
        c["grads"]["dL_dmeans3D"] = c["grads"]["dL_dmeans3D"] * 1.05
    elif what == "grad_one_component":
        c["grads"]["dL_drotations"][:, 2] *= 0.9
    elif what == "image_offset":
        c["color"][1] += 3e-4
    elif what == "index_shift":
        nc = c["n_contrib"].copy(); nc[0][40:44, 50:90] += 1; c["n_contrib"] = nc
    else:
        c["grads"]["dL_dopacity"] = c["grads"]["dL_dopacity"] * 0.97
    with pytest.raises(AssertionError):
        pt.check_case(variant, "precomp", c, f32, fma, truth)


def test_floor_run_is_the_exact_blend_of_the_float32_geometry(case):
    """truth["floor"] (oracle.Truth(f32_geometry=True)): float64 blend of the float32 run's per-gaussian state.  It is a different computation from the
    truth (its inputs were rounded), it is what a perfect float32-state blend would return -- so the criterion accepts it with zero distance from itself --
    and for EWA / PLANE, whose blend inputs are well conditioned, it lies within float32 accuracy of the truth."""
    variant, f32, fma, truth, ints = case
    flo = truth["floor"]
    d = pt._rel(flo["grads"]["dL_dmeans3D"], truth["grads"]["dL_dmeans3D"])
    assert 0.0 < d < (2e-2 if variant == "surfel" else 1e-3)
    robust = truth["margin"] > 1.0
    rows = pt.robust_rows(truth["splat"], ~robust, d and truth["grads"]["dL_dmeans3D"].shape[0])
    for k in ("dL_dmeans3D", "dL_drotations", "dL_dopacity"):
        rep = {}
        try:
            pt.check_grad(k, flo["grads"][k], [f32["grads"][k], fma["grads"][k]], truth["grads"][k], rows, report=rep, floor=flo["grads"][k])
        except AssertionError as e:      # (its gate decisions on the FRAGILE rows may legitimately differ from the float32 oracle's: only that bar may trip)
            assert "flipping splats" in str(e)
        assert rep[k]["rel_l2_vs_floor"] == 0.0 and rep[k]["floor_rel_l2"] == rep[k]["rel_l2"]
    rep = pt.check_map("color", flo["color"], [f32["color"], fma["color"]], truth["color"], robust, floor=flo["color"])
    assert rep["robust_px_beyond_tol_vs_floor"] == 0 and rep["floor_robust_px_beyond_tol"] == rep["robust_px_beyond_tol"]


def test_criterion_rejects_a_blend_error_above_the_nominal_tolerance_even_below_the_floor(case):
    """The bar against the floor run has no relative term: a 0.3 % error in a surfel's geometric gradient is rejected although the float32 floor of that
    tensor (and the unfused float32 oracle's own error) may be larger."""
    variant, f32, fma, truth, ints = case
    flo = truth["floor"]
    rows = pt.robust_rows(truth["splat"], truth["margin"] <= 1.0, flo["grads"]["dL_drotations"].shape[0])
    with pytest.raises(AssertionError, match="float32.geometry"):
        pt.check_grad("dL_drotations", flo["grads"]["dL_drotations"] * 1.003, [f32["grads"]["dL_drotations"], fma["grads"]["dL_drotations"]],
                      truth["grads"]["dL_drotations"], rows, floor=flo["grads"]["dL_drotations"])
