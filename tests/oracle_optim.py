"""numpy restatement of the Adam update the reference's models are stepped with (torch.optim.Adam, no amsgrad / weight decay; the arithmetic of
torch's single-tensor implementation).  TEST INFRASTRUCTURE ONLY.  tests/test_optim_cpu.py pins it against torch.optim.Adam itself."""
import math

import numpy as np


def adam_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, lr_scale=None):
    """One step (t = 1, 2, ...) in float32; returns the new (p, m, v)."""
    f = np.float32
    p, g, m, v = (np.asarray(a, np.float32) for a in (p, g, m, v))
    m = m + (g - m) * f(1.0 - beta1)
    v = v * f(beta2) + g * g * f(1.0 - beta2)
    step_size = f(lr / (1.0 - beta1 ** t))
    inv_bc2_sqrt = f(1.0) / f(math.sqrt(1.0 - beta2 ** t))
    denom = np.sqrt(v) * inv_bc2_sqrt + f(eps)
    sc = f(1.0) if lr_scale is None else np.asarray(lr_scale, np.float32)
    p = p - (step_size * sc) * (m / denom)
    return p.astype(np.float32), m.astype(np.float32), v.astype(np.float32)
