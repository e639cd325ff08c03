"""GPU parity of the fused L1+SSIM loss (gsrast.losses.l1_ssim -> gsr_loss_l1_ssim) vs the oracle and, at 1080p, the torch chain."""
import numpy as np
import pytest
import torch

import oracle
import ref_loss_torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(shape, seed, noise=0.15):
    r = np.random.default_rng(seed)
    gt = r.uniform(0, 1, shape).astype(np.float32)
    return np.clip(gt + r.normal(0, noise, shape), 0, 1).astype(np.float32), gt


@pytest.mark.parametrize("shape,lam", [((3, 37, 53), 0.2), ((1, 8, 9), 0.5), ((3, 64, 48), 1.0), ((2, 11, 30), 0.0), ((3, 16, 16), 0.2),
                                       ((3, 1, 1), 0.2), ((3, 200, 333), 0.2)])
def test_l1_ssim_matches_oracle(shape, lam):
    from gsrast.losses import l1_ssim
    img, gt = _pair(shape, sum(shape))
    out, d = oracle.loss_l1_ssim(img, gt, lam)
    x = torch.tensor(img, device=DEV, requires_grad=True)
    loss, parts = l1_ssim(x, torch.tensor(gt, device=DEV), lam, return_parts=True)
    (loss * 3.0).backward()
    np.testing.assert_allclose([parts[0].item(), parts[1].item(), loss.item()], out, rtol=2e-5, atol=2e-6)
    g = x.grad.cpu().numpy() / 3.0
    assert np.abs(g - d).max() <= 1e-4 * np.abs(d).max() + 1e-9


def test_l1_ssim_full_hd_vs_torch_chain():
    from gsrast.losses import l1_ssim
    img, gt = _pair((3, 1080, 1920), 5)
    x = torch.tensor(img, device=DEV, requires_grad=True)
    y = torch.tensor(gt, device=DEV)
    loss = l1_ssim(x, y, 0.2)
    loss.backward()
    xr = torch.tensor(img, device=DEV, requires_grad=True)
    Lr, _, _ = ref_loss_torch.loss(xr.unsqueeze(0), y.unsqueeze(0), 0.2)
    Lr.backward()
    assert abs(loss.item() - Lr.item()) < 1e-5
    g, gr = x.grad, xr.grad
    assert (g - gr).abs().max().item() <= 2e-3 * gr.abs().max().item()
    assert ((g - gr).norm() / gr.norm()).item() < 1e-4
    # identical images: SSIM = 1, loss = 0
    l0, parts = l1_ssim(y, y, 0.2, return_parts=True)
    assert abs(parts[1].item() - 1.0) < 1e-6 and abs(l0.item()) < 1e-6
