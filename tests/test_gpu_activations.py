"""GPU test (pytest -m gpu): gsrast.activations.gaussian_activations (include/gsrast.h gsr_gauss_activations[_backward]) against the reference's own
formulation -- torch.exp / torch.nn.functional.normalize / torch.sigmoid on the three parameter tensors
(/root/reference/gssr/gaussian/vanilla_gaussian.py:86-90,250-269) -- values and gradients, incl. a zero quaternion (normalize's eps clamp), an output
the loss never touches (null upstream gradient) and the 2-axis scaling of the surfel models."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,S", [(1, 3), (257, 2), (100003, 3)])
def test_gaussian_activations_match_torch(P, S):
    from gsrast.activations import gaussian_activations
    g = torch.Generator().manual_seed(P)
    s0 = (torch.randn(P, S, generator=g) * 1.5 - 3.0).cuda()
    q0 = torch.randn(P, 4, generator=g).cuda()
    o0 = (torch.randn(P, 1, generator=g) * 2.0).cuda()
    if P > 3:
        q0[3] = 0.0                                     # F.normalize: x / max(|x|, 1e-12) -> 0, gradient g / 1e-12
        q0[2] *= 1e-3
    w = [torch.randn(P, S, generator=g).cuda(), torch.randn(P, 4, generator=g).cuda(), torch.randn(P, 1, generator=g).cuda()]

    def run(fn, use=(True, True, True)):
        leaves = [t.clone().requires_grad_(True) for t in (s0, q0, o0)]
        outs = fn(*leaves)
        loss = sum((o * ww).sum() for o, ww, u in zip(outs, w, use) if u)
        loss.backward()
        return [o.detach() for o in outs], [l.grad for l in leaves]
    ref = lambda s, q, o: (torch.exp(s), torch.nn.functional.normalize(q), torch.sigmoid(o))
    for use in ((True, True, True), (True, False, True), (False, True, False)):
        (a, ga), (b, gb) = run(gaussian_activations, use), run(ref, use)
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7), (x - y).abs().max().item()
        for k, (x, y) in enumerate(zip(ga, gb)):
            if y is None:
                assert x is None or float(x.abs().max()) == 0.0
                continue
            ok = torch.isfinite(y).all(dim=-1) if y.dim() > 1 else torch.isfinite(y)
            assert torch.allclose(x[ok], y[ok], rtol=1e-5, atol=1e-6 * float(y[ok].abs().max() + 1)), (k, (x[ok] - y[ok]).abs().max().item())
