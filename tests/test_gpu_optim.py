"""GPU: gsrast.optim.Adam (one fused HIP kernel per parameter tensor, include/gsrast.h gsr_adam_step) against the numpy oracle and against
torch.optim.Adam on the same parameters -- including the per-group learning-rate rewrites and the optimizer-state surgery the reference's
densification performs (gssr/gaussian/vanilla_gaussian.py: cat_tensors_to_optimizer / _prune_optimizer)."""
import numpy as np
import pytest
import torch

import oracle_optim

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _models(seed, shapes, lrs, cls):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    return ps, cls([{"params": [p], "lr": lr, "name": f"g{i}"} for i, (p, lr) in enumerate(zip(ps, lrs))], lr=0.0, eps=1e-15)


def test_fused_adam_matches_oracle_and_torch():
    from gsrast.optim import Adam
    shapes = [(30001, 3), (30001, 1), (30001, 15, 3), (7,), (1,)]
    lrs = [1.6e-4, 5e-2, 1.25e-4, 1e-3, 1e-2]
    pa, oa = _models(3, shapes, lrs, Adam)
    pb, ob = _models(3, shapes, lrs, lambda groups, **kw: torch.optim.Adam(groups, foreach=False, **kw))
    st = [(p.detach().cpu().numpy().copy(), np.zeros(s, np.float32), np.zeros(s, np.float32)) for p, s in zip(pa, shapes)]
    g = torch.Generator().manual_seed(9)
    for t in range(1, 7):
        if t == 4:                                                   # the reference's schedulers rewrite param_group['lr'] every iteration
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3.0e-5
            lrs[0] = 3.0e-5
        grads = [torch.randn(s, generator=g) * (0.1 if t % 2 else 3.0) for s in shapes]
        for ps in (pa, pb):
            for p, gr in zip(ps, grads):
                p.grad = gr.to(DEV)
        oa.step(); ob.step()
        oa.zero_grad(set_to_none=True); ob.zero_grad(set_to_none=True)
        for i, (gr, lr) in enumerate(zip(grads, lrs)):
            st[i] = oracle_optim.adam_step(st[i][0], gr.numpy(), st[i][1], st[i][2], t, lr, eps=1e-15)
            sa = oa.state[pa[i]]
            assert set(sa.keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sa["step"]) == t
            assert np.allclose(sa["exp_avg"].cpu().numpy(), st[i][1], rtol=1e-5, atol=2e-6)
            assert np.allclose(sa["exp_avg_sq"].cpu().numpy(), st[i][2], rtol=1e-5, atol=1e-12)
            assert np.allclose(pa[i].detach().cpu().numpy(), st[i][0], rtol=0, atol=1e-4 * lr + 5e-7), (t, i)
            assert np.allclose(pa[i].detach().cpu().numpy(), pb[i].detach().cpu().numpy(), rtol=0, atol=1e-4 * lr + 5e-7), (t, i)
    assert oa.state_dict()["param_groups"][0]["name"] == "g0" and len(oa.state_dict()["state"]) == len(shapes)


def test_state_surgery_of_densification_and_uncovered_parameters():
    """cat_tensors_to_optimizer-style surgery (new Parameter, concatenated moments under the new key) keeps working, a parameter without
    gradient is skipped, and a float64 parameter goes through torch's own update."""
    from gsrast.optim import Adam
    g = torch.Generator().manual_seed(2)
    p = torch.nn.Parameter(torch.randn(100, 3, generator=g).to(DEV)); q = torch.nn.Parameter(torch.randn(5, generator=g).double().to(DEV))
    r = torch.nn.Parameter(torch.randn(4, generator=g).to(DEV))
    opt = Adam([{"params": [p], "lr": 1e-2, "name": "xyz"}, {"params": [q], "lr": 1e-2, "name": "f64"}, {"params": [r], "lr": 1e-2, "name": "nograd"}],
               lr=0.0, eps=1e-15)
    p.grad = torch.ones_like(p); q.grad = torch.ones_like(q)
    r0 = r.detach().clone(); q0 = q.detach().clone()
    opt.step()
    assert torch.equal(r, r0) and len(opt.state[r]) == 0
    assert torch.allclose(q, q0 - 1e-2, atol=1e-9) and opt.state[q]["exp_avg"].dtype == torch.float64
    group = opt.param_groups[0]
    stored = opt.state.pop(group["params"][0])
    ext = torch.randn(20, 3, generator=g).to(DEV)
    stored["exp_avg"] = torch.cat((stored["exp_avg"], torch.zeros_like(ext)), dim=0)
    stored["exp_avg_sq"] = torch.cat((stored["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
    group["params"][0] = torch.nn.Parameter(torch.cat((p.detach(), ext), dim=0).requires_grad_(True))
    opt.state[group["params"][0]] = stored
    newp = group["params"][0]
    before = newp.detach().clone()
    newp.grad = torch.ones_like(newp)
    opt.step()
    assert float(opt.state[newp]["step"]) == 2.0 and opt.state[newp]["exp_avg"].shape == (120, 3)
    d = before - newp.detach()
    assert torch.all(d > 0) and torch.allclose(d[:100], d[:1].expand(100, 3), atol=1e-7)


def test_more_tensors_than_one_table_with_empty_ones():
    """30 parameter tensors, some with numel() == 0 (features_rest is (N, 0, 3) at sh_degree 0), on one device: more than the 24 slots of
    one launch table.  ADVICE r2: the batch loop used to restart 24 indices further whatever it had consumed, so an empty tensor inside a
    batch made the next batch re-apply entries -- a silent double step.  Every tensor must get exactly one torch-identical update."""
    from gsrast.optim import Adam
    shapes = [(5000 + 37 * i, 3) if i % 7 != 2 else (11, 0, 3) for i in range(30)]
    lrs = [1e-3 * (1 + i % 5) for i in range(30)]
    pa, oa = _models(5, shapes, lrs, Adam)
    pb, ob = _models(5, shapes, lrs, lambda groups, **kw: torch.optim.Adam(groups, foreach=False, **kw))
    g = torch.Generator().manual_seed(1)
    for t in range(1, 4):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for ps in (pa, pb):
            for p, gr in zip(ps, grads):
                p.grad = gr.to(DEV)
        oa.step(); ob.step()
        for i in range(30):
            assert float(oa.state[pa[i]]["step"]) == t
            assert torch.allclose(pa[i], pb[i], rtol=0, atol=1e-4 * lrs[i] + 5e-7), (t, i)
            assert torch.allclose(oa.state[pa[i]]["exp_avg"], ob.state[pb[i]]["exp_avg"], rtol=1e-5, atol=2e-6), (t, i)
            assert torch.allclose(oa.state[pa[i]]["exp_avg_sq"], ob.state[pb[i]]["exp_avg_sq"], rtol=1e-5, atol=1e-12), (t, i)


def test_step_recorded_in_a_hip_graph_matches_torch():
    """step() issued during stream capture records ONE launch that reads step_size / bias correction from a device buffer; prepare_replay()
    refreshes it (and honours a learning-rate rewrite) before every replay.  Six replays == six steps of torch.optim.Adam on the same gradients."""
    from gsrast.optim import Adam
    shapes = [(20001, 3), (20001, 1), (7,), (513, 4)]
    lrs = [1.6e-4, 5e-2, 1e-3, 2e-3]
    pa, oa = _models(7, shapes, lrs, Adam)
    pb, ob = _models(7, shapes, lrs, lambda groups, **kw: torch.optim.Adam(groups, foreach=False, **kw))
    gen = torch.Generator().manual_seed(3)
    static_g = [torch.zeros(s, device=DEV) for s in shapes]
    for p, g in zip(pa, static_g):
        p.grad = g                                   # static gradient slots, as a recorded backward leaves them
    def feed():
        gs = [torch.randn(s, generator=gen) for s in shapes]
        for sg, g, q in zip(static_g, gs, pb):
            sg.copy_(g.to(DEV)); q.grad = g.to(DEV)
    feed(); oa.step(); ob.step()                     # one eager step creates the state
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        oa.step()
    for t in range(2, 8):
        if t == 5:
            oa.param_groups[0]["lr"] = 3e-5; ob.param_groups[0]["lr"] = 3e-5
        feed()
        oa.prepare_replay(); graph.replay(); ob.step()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert float(oa.state[a]["step"]) == t
            lr = oa.param_groups[i]["lr"]
            assert torch.allclose(a, b, rtol=0, atol=1e-4 * lr + 5e-7), (t, i, (a - b).abs().max().item())


def test_two_graphs_over_one_optimizer_keep_their_own_scalars():
    """Two recorded step() calls over ONE optimizer that cover different parameter subsets (say one graph per image size, one of which never
    reaches some tensors): every capture owns its hyper-parameter buffer and its list of captured tensors, and prepare_replay(handles=...) moves
    only the step counters of the graph about to be replayed (the second capture used to overwrite the first one's bookkeeping, so the
    older graph replayed with another tensor's step size)."""
    from gsrast.optim import Adam
    shapes = [(4001, 3), (4001, 1), (9,), (257, 4)]
    lrs = [1.6e-4, 5e-2, 1e-3, 2e-3]
    pa, oa = _models(11, shapes, lrs, Adam)
    pb, ob = _models(11, shapes, lrs, lambda groups, **kw: torch.optim.Adam(groups, foreach=False, **kw))
    gen = torch.Generator().manual_seed(5)
    static_g = [torch.zeros(s, device=DEV) for s in shapes]
    subsets = {"A": [0, 1, 2, 3], "B": [3, 1]}

    def feed(which):
        for i in range(len(shapes)):
            pa[i].grad = None; pb[i].grad = None
        for i in subsets[which]:
            g = torch.randn(shapes[i], generator=gen).to(DEV)
            static_g[i].copy_(g); pa[i].grad = static_g[i]; pb[i].grad = g
    graphs, handles = {}, {}
    for which in ("A", "B"):
        feed(which); oa.step(); ob.step()            # an eager step before EACH capture (state + a spare hyper buffer)
        torch.cuda.synchronize()
        n0 = len(oa.captured_steps())
        graphs[which] = torch.cuda.CUDAGraph()
        feed(which)                                  # the gradient slots the recorded launch reads
        with torch.cuda.graph(graphs[which]):
            oa.step()
        handles[which] = oa.captured_steps()[n0:]
        assert len(handles[which]) == 1 and len(handles[which][0][0]) == len(subsets[which])
    assert handles["A"][0][1].data_ptr() != handles["B"][0][1].data_ptr()       # separate device buffers
    for which in ("A", "B", "B", "A", "A", "B"):
        feed(which)
        oa.prepare_replay(handles=handles[which]); graphs[which].replay(); ob.step()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert float(oa.state[a]["step"]) == float(ob.state[b]["step"]), (which, i)
            lr = oa.param_groups[i]["lr"]
            assert torch.allclose(a, b, rtol=0, atol=1e-4 * lr + 5e-7), (which, i, (a - b).abs().max().item())


def test_shadow_parameters_two_passes_one_update():
    """Two passes over the same parameters in one backward (PGSR's reference + neighbour camera): with shadow leaves for the second pass and
    Adam.add_shadows the update kernel reads both gradients; parameters and optimizer state stay bit-identical to the plain form in which
    autograd adds the two passes' gradients itself, over several steps, for tensors, a module, an odd-sized tensor and a parameter only the second
    pass touches."""
    from gsrast.optim import Adam, shadow_parameters
    torch.manual_seed(5)

    def model():
        g = torch.Generator().manual_seed(11)
        w = torch.randn(4099, 3, generator=g).to(DEV).requires_grad_(True)
        b = torch.randn(7, generator=g).to(DEV).requires_grad_(True)
        only2 = torch.randn(33, generator=g).to(DEV).requires_grad_(True)
        lin = torch.nn.Linear(3, 5).to(DEV)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(5, 3, generator=g)); lin.bias.copy_(torch.randn(5, generator=g))
        return {"w": w, "b": b, "only2": only2, "lin": lin}

    def loss_of(L1, L2, x1, x2):
        a = (L1["lin"](x1 * L1["w"]).tanh().sum(1) * 0.01).sum() + (L1["b"] ** 2).sum()
        c = (L2["lin"](x2 * L2["w"]).sin().sum(1) * 0.02).sum() + (L2["b"] * 0.5).sum() + (L2["only2"] ** 3).sum()
        return a + c

    A, B = model(), model()
    pa = [A["w"], A["b"], A["only2"]] + list(A["lin"].parameters())
    pb = [B["w"], B["b"], B["only2"]] + list(B["lin"].parameters())
    oa, ob = Adam(pa, lr=1e-2, eps=1e-15), Adam(pb, lr=1e-2, eps=1e-15)
    S = shadow_parameters(B)
    assert S["w"].data_ptr() == B["w"].data_ptr() and S["lin"].weight.data_ptr() == B["lin"].weight.data_ptr() and S["w"] is not B["w"]
    ob.add_shadows(B, S)
    g = torch.Generator().manual_seed(2)
    for t in range(5):
        x1 = torch.randn(4099, 3, generator=g).to(DEV); x2 = torch.randn(4099, 3, generator=g).to(DEV)
        loss_of(A, A, x1, x2).backward()
        loss_of(B, S, x1, x2).backward()
        assert B["only2"].grad is None and S["only2"].grad is not None
        oa.step(); ob.step()
        oa.zero_grad(set_to_none=True); ob.zero_grad(set_to_none=True)
        assert S["w"].grad is None and S["lin"].weight.grad is None
        for p, q in zip(pa, pb):
            assert torch.equal(p, q), t
            assert torch.equal(oa.state[p]["exp_avg"], ob.state[q]["exp_avg"]) and torch.equal(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"])
        assert torch.equal(S["w"], B["w"])                       # the shadow sees the in-place update
