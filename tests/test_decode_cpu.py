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


@pytest.mark.parametrize("mode", ["floor", "round", "ceil", "progressive"])
def test_lod_mask_oracle_matches_torch_transcription(mode):
    """refd_lod_mask vs the reference's torch ops (octree_gaussian.py:184-203,255-267) on CPU float32."""
    import torch
    r = np.random.default_rng(3)
    Na, levels, fork, vs, sd = 4000, 6, 2.0, 0.4, 12.0
    anchor = r.uniform(-8, 8, (Na, 3)).astype(np.float32); level = r.integers(0, levels, Na).astype(np.int32)
    extra = r.uniform(-0.3, 0.3, Na).astype(np.float32); cam = np.array([0.5, -1.0, 3.0], np.float32)
    m, pr, tr = oracle_decode.lod_mask(anchor, level, extra, cam, vs, fork, sd, 1.0, levels, ["floor", "round", "ceil", "progressive"].index(mode))
    A, Lv, Ex = torch.tensor(anchor), torch.tensor(level).unsqueeze(1), torch.tensor(extra)
    anchor_pos = A + (vs / 2) / (float(fork) ** Lv)
    dist = torch.sqrt(torch.sum((anchor_pos - torch.tensor(cam)) ** 2, dim=1)) * 1.0
    pred = torch.log2(sd / dist) / np.log2(fork) + Ex
    cur = levels - 1
    if mode == "progressive":
        p2 = torch.clamp(pred + 1.0, min=0.9999, max=cur + 0.9999); il = torch.floor(p2).int()
        near = (p2 - torch.round(p2)).abs() < 1e-4
        assert np.allclose(pr[~near.numpy()], torch.frac(p2).numpy()[~near.numpy()], atol=1e-4)
        assert np.array_equal(tr[~near.numpy()], (Lv.squeeze(1) == il).numpy()[~near.numpy()])
    else:
        f = {"floor": torch.floor, "round": torch.round, "ceil": torch.ceil}[mode]
        il = torch.clamp(f(pred).int(), min=0, max=cur)
        tgt = pred - (0.5 if mode == "round" else 0.0)
        near = (tgt - torch.round(tgt)).abs() < 1e-4
    ref = (Lv.squeeze(1) <= il).numpy()
    assert np.array_equal(m[~near.numpy()], ref[~near.numpy()]) and 0.05 < m.mean() < 0.95
