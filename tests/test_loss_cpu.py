"""Photometric-loss oracle (oracle/gsl_oracle.c) vs the reference's torch formula under float64 autograd."""
import numpy as np
import pytest
import torch

import oracle
import ref_loss_torch


@pytest.mark.parametrize("shape,lam", [((3, 37, 53), 0.2), ((1, 8, 9), 0.5), ((3, 64, 48), 1.0), ((2, 11, 30), 0.0)])
def test_l1_ssim_oracle_matches_autograd(shape, lam):
    r = np.random.default_rng(sum(shape))
    gt = r.uniform(0, 1, shape).astype(np.float32)
    img = np.clip(gt + r.normal(0, 0.15, shape), 0, 1).astype(np.float32)
    out, d = oracle.loss_l1_ssim(img, gt, lam)
    x = torch.tensor(img, dtype=torch.float64, requires_grad=True)
    L, l1, s = ref_loss_torch.loss(x, torch.tensor(gt, dtype=torch.float64), lam)
    L.backward()
    np.testing.assert_allclose(out, [l1.item(), s.item(), L.item()], rtol=2e-6, atol=2e-7)
    g = x.grad.numpy()
    assert np.abs(d - g).max() <= 2e-5 * np.abs(g).max() + 1e-10


def test_ssim_identity_and_symmetry():
    r = np.random.default_rng(0)
    a = r.uniform(0, 1, (3, 20, 24)).astype(np.float32)
    out, d = oracle.loss_l1_ssim(a, a, 0.2)
    assert abs(out[1] - 1.0) < 1e-6 and abs(out[2]) < 1e-6 and np.abs(d).max() < 1e-6
    b = r.uniform(0, 1, (3, 20, 24)).astype(np.float32)
    assert abs(oracle.loss_l1_ssim(a, b, 1.0)[0][1] - oracle.loss_l1_ssim(b, a, 1.0)[0][1]) < 1e-6
