"""Oracle (oracle/gsd_oracle.c) vs the float64 torch transcription of the reference's generate_neural_gaussians."""
import numpy as np
import pytest

import decode_cases
import oracle_decode
import ref_decode_torch

CASES = [dict(), dict(A=0, k=5), dict(dist_o=True, dist_c=True, dist_k=True, seed=1), dict(level=True, progressive=True, seed=2),
         dict(dist_k=True, level=True, A=16, k=12, seed=3)]


@pytest.mark.parametrize("kw", CASES)
def test_decode_oracle_matches_autograd(kw):
    case = decode_cases.make_case(Na=300, **kw)
    o = oracle_decode.forward(case)
    ref, leaves = ref_decode_torch.decode(case, mask_override=o["mask"])
    # the gate itself: wherever the float64 value is clearly non-zero the masks agree
    no64 = ref["neural_opacity"].detach().numpy()
    clear = np.abs(no64) > 1e-6
    assert np.array_equal((no64 > 0)[clear], o["mask"].astype(bool)[clear])
    assert 0 < o["P"] < o["mask"].size
    for n in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity"):
        np.testing.assert_allclose(o[n], ref[n].detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=n)
    dL = decode_cases.make_out_grads(o["P"], seed=kw.get("seed", 0))
    g = oracle_decode.backward(case, o["mask"], dL)
    g64 = ref_decode_torch.backward(ref, leaves, dL)
    for n, v in g.items():
        r = g64[n]
        scale = np.abs(r).max() + 1e-12
        assert np.abs(v - r).max() / scale < 2e-5, (n, np.abs(v - r).max(), scale)


def test_decode_oracle_compaction_order_and_invisible_rows():
    case = decode_cases.make_case(Na=200, seed=5)
    o = oracle_decode.forward(case)
    k = case["k"]
    rows = np.nonzero(o["mask"])[0]
    a = case["vis_idx"][rows // k]; j = rows % k
    np.testing.assert_allclose(o["xyz"], case["anchor"][a] + case["offset"][a, j] * case["scaling"][a, :3], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(o["rot"], axis=1), 1.0, atol=1e-5)
    g = oracle_decode.backward(case, o["mask"], decode_cases.make_out_grads(o["P"]))
    hidden = np.setdiff1d(np.arange(200), case["vis_idx"])
    assert hidden.size and not g["feat"][hidden].any() and not g["anchor"][hidden].any() and not g["offset"][hidden].any()
