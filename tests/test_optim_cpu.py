"""The Adam oracle (tests/oracle_optim.py) against torch.optim.Adam itself -- the optimizer the reference constructs
(gssr/gaussian/vanilla_gaussian.py:133: Adam(l, lr=0.0, eps=1e-15), per-group learning rates) -- over several steps."""
import numpy as np
import pytest
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


def test_uncovered_parameters_take_torchs_update_and_hooks_fire_once():
    """Host tensors are not covered by the HIP kernel: gsrast.optim.Adam must give them exactly torch.optim.Adam's update through torch's
    functional form -- and the optimizer's step hooks must fire once per step(), not twice (the fallback used to call
    super().step() from inside the overridden step())."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gs-sr_amd"))
    import torch
    from gsrast.optim import Adam
    torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))])          # a plain Adam in the process: torch wraps Adam.step with its profile hook
    g = torch.Generator().manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(7, 3, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g).double())]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = Adam([{"params": [a[0]], "lr": 1e-2}, {"params": [a[1]], "lr": 3e-3}], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [b[0]], "lr": 1e-2}, {"params": [b[1]], "lr": 3e-3}], lr=0.0, eps=1e-15, foreach=False)
    fired = []
    oa.register_step_post_hook(lambda *args: fired.append(1))
    for t in range(3):
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g).to(p.dtype)
            p.grad = gr.clone(); q.grad = gr.clone()
        oa.step(); ob.step()
        for p, q in zip(a, b):
            assert torch.equal(p, q) and torch.equal(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"]) and float(oa.state[p]["step"]) == t + 1
    assert len(fired) == 3
    assert len(oa.param_groups) == 2 and oa.param_groups[0]["params"][0] is a[0]


def test_shadow_parameters_on_the_torch_path():
    """CPU parameters are not covered by the HIP kernel: the shadows' gradients are added before torch's own update (same result as one set of
    leaves), the shadows share storage and are cleared by zero_grad."""
    from gsrast.optim import Adam, shadow_parameters
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(50, 3, generator=g)
    lin0 = torch.nn.Linear(3, 2)
    A = {"w": w0.clone().requires_grad_(True), "lin": torch.nn.Linear(3, 2)}
    B = {"w": w0.clone().requires_grad_(True), "lin": torch.nn.Linear(3, 2)}
    for M in (A, B):
        M["lin"].load_state_dict(lin0.state_dict())
    oa = Adam([A["w"]] + list(A["lin"].parameters()), lr=1e-2)
    ob = Adam([B["w"]] + list(B["lin"].parameters()), lr=1e-2)
    S = shadow_parameters(B)
    assert S["w"].data_ptr() == B["w"].data_ptr() and S["lin"].weight.data_ptr() == B["lin"].weight.data_ptr()
    ob.add_shadows(B, S)
    with pytest.raises(ValueError):
        ob.add_shadows(B, {"w": w0.clone().requires_grad_(True), "lin": torch.nn.Linear(3, 2)})       # not the same storage
    for t in range(3):
        x1, x2 = torch.randn(50, 3, generator=g), torch.randn(50, 3, generator=g)
        (A["lin"](x1 * A["w"]).sum() + (A["lin"](x2 * A["w"]) ** 2).sum()).backward()
        (B["lin"](x1 * B["w"]).sum() + (S["lin"](x2 * S["w"]) ** 2).sum()).backward()
        oa.step(); ob.step(); oa.zero_grad(); ob.zero_grad()
        assert S["w"].grad is None
        assert torch.allclose(A["w"], B["w"], rtol=0, atol=1e-7) and torch.allclose(A["lin"].weight, B["lin"].weight, rtol=0, atol=1e-7)
