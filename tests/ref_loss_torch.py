"""Independent torch restatement of the photometric loss the reference trains with -- (1 - lam) * L1 + lam * (1 - SSIM) with an
11 x 11, sigma 1.5 Gaussian window and zero padding (definition: gssr/scene/vanilla_scene.py:29-69).  TEST INFRASTRUCTURE ONLY.

Written from the formula, not from the reference's code: the window is applied as two 1-D passes (rows, then columns) through
`unfold`-free strided sums, the five local moments are stacked into one tensor so a single pair of passes filters them all, and SSIM
is evaluated per pixel as
        (2 m_x m_y + c1) (2 cov_xy + c2) / ((m_x^2 + m_y^2 + c1) (var_x + var_y + c2)),   c1 = 0.01^2, c2 = 0.03^2.
tests/test_loss_cpu.py holds it (float64) against the reference-run fixture ref_loss_l1_ssim.npz and the C oracle against both."""
import torch

WIN, SIGMA = 11, 1.5
C1, C2 = 0.01 ** 2, 0.03 ** 2


def _taps(dtype, device):
    x = torch.arange(WIN, dtype=torch.float64) - WIN // 2
    g = torch.exp(-(x * x) / (2.0 * SIGMA * SIGMA))
    return (g / g.sum()).to(dtype=dtype, device=device)


def _blur(t):
    """Separable Gaussian filter over the last two dims with zero padding; t is (..., H, W)."""
    g = _taps(t.dtype, t.device)
    r = WIN // 2
    H, W = t.shape[-2], t.shape[-1]
    p = torch.nn.functional.pad(t, (r, r, r, r))
    rows = sum(g[k] * p[..., :, k:k + W] for k in range(WIN))        # along x
    return sum(g[k] * rows[..., k:k + H, :] for k in range(WIN))     # along y


def ssim(x, y):
    m = _blur(torch.stack([x, y, x * x, y * y, x * y]))
    mx, my, mxx, myy, mxy = m[0], m[1], m[2], m[3], m[4]
    vx, vy, cxy = mxx - mx * mx, myy - my * my, mxy - mx * my
    num = (2.0 * mx * my + C1) * (2.0 * cxy + C2)
    den = (mx * mx + my * my + C1) * (vx + vy + C2)
    return (num / den).mean()


def loss(img, gt, lambda_dssim):
    """-> (total, l1, ssim); img / gt are (..., C, H, W) of any float dtype."""
    l1 = (img - gt).abs().mean()
    s = ssim(img, gt)
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - s), l1, s
