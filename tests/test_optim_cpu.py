"""The Adam oracle (tests/oracle_optim.py) against torch.optim.Adam itself -- the optimizer the reference constructs
(gssr/gaussian/vanilla_gaussian.py:133: Adam(l, lr=0.0, eps=1e-15), per-group learning rates) -- over several steps."""
import numpy as np
import torch

import oracle_optim


def test_oracle_matches_torch_adam_over_steps():
    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 1), (257,), (64, 16, 3)]
    lrs = [1.6e-4, 5e-2, 1e-3, 2.5e-3]
    ps = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)], lr=0.0, eps=1e-15, foreach=False)
    state = [(p.detach().numpy().copy(), np.zeros(s, np.float32), np.zeros(s, np.float32)) for p, s in zip(ps, shapes)]
    for t in range(1, 8):
        grads = [torch.randn(s, generator=g) * (10.0 ** (t % 3 - 1)) for s in shapes]
        for p, gr in zip(ps, grads):
            p.grad = gr.clone()
        opt.step()
        for i, (gr, lr) in enumerate(zip(grads, lrs)):
            state[i] = oracle_optim.adam_step(state[i][0], gr.numpy(), state[i][1], state[i][2], t, lr, eps=1e-15)
            st = opt.state[ps[i]]
            assert np.allclose(state[i][1], st["exp_avg"].numpy(), rtol=1e-5, atol=2e-6)        # torch's lerp_ rounds differently by an ulp
            assert np.allclose(state[i][2], st["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-12)
            assert np.allclose(state[i][0], ps[i].detach().numpy(), rtol=0, atol=1e-4 * lr + 5e-7), (t, i)      # a step is ~lr; the parameters are O(1) floats


def test_lr_scale_is_a_per_element_learning_rate():
    r = np.random.default_rng(1)
    p, g = r.normal(size=50).astype(np.float32), r.normal(size=50).astype(np.float32)
    z = np.zeros(50, np.float32)
    a = oracle_optim.adam_step(p, g, z, z, 1, 1.0, lr_scale=np.full(50, 3e-3, np.float32))[0]
    b = oracle_optim.adam_step(p, g, z, z, 1, 3e-3)[0]
    assert np.allclose(a, b, rtol=0, atol=1e-6)
