"""GPU: the round-3 entry points of the C ABI -- gsr_backward_ex (accumulator scratch left zeroed by the preprocess backward and reused without
a memset) and gsr_forward_async (no host synchronisation: the form a HIP graph records) -- against the established paths."""
import numpy as np
import pytest
import torch

import scenes

pytestmark = pytest.mark.gpu


def _run(hr, variant, sc, og):
    r = hr.run(variant, sc, og)
    return r


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_accumulator_scratch_is_left_zero_and_reused(variant):
    """Three backward calls in a row through the cached, self-cleaning scratch (GSR_BWD_SCRATCH_IS_ZERO | GSR_BWD_LEAVE_ZERO) give the gradients
    of the memset path: a row the preprocess backward failed to clear would leak the previous call's gradient into the next."""
    import hiprun as hr
    from gsrast import rasterize as rz
    W, H, P = 240, 160, 5000
    sc = scenes.make_scene(variant, P, W, H, seed=5)
    ogs = [scenes.random_out_grads(variant, W, H, seed=s, scale=1.0) for s in (1, 2, 3)]
    assert rz._ACC_REUSE
    rz._ACC_CACHE.clear()
    got = [hr.run(variant, sc, og)["grads"] for og in ogs]
    assert len(rz._ACC_CACHE) == 1
    for t in rz._ACC_CACHE.values():
        assert not bool(t.any()), "the scratch must hold only zeros between calls"
    rz._ACC_REUSE = False
    try:
        ref = [hr.run(variant, sc, og)["grads"] for og in ogs]
    finally:
        rz._ACC_REUSE = True
    for g, r in zip(got, ref):
        for k in r:
            if r[k] is None:
                continue
            d = np.linalg.norm(g[k].astype(np.float64) - r[k]); n = np.linalg.norm(r[k].astype(np.float64))
            assert d <= 2e-5 * n + 1e-30, (variant, k, d, n)          # the two paths differ by the order of the float atomics only


@pytest.mark.parametrize("variant", ["ewa", "surfel", "plane"])
def test_forward_async_matches_and_reports_overflow(variant):
    import hiprun as hr
    from gsrast import rasterize as rz
    W, H, P = 320, 200, 8000
    sc = scenes.make_scene(variant, P, W, H, seed=9)
    og = scenes.random_out_grads(variant, W, H, seed=9, scale=1.0)
    ref = hr.run(variant, sc, og)
    R = hr.run_raw(variant, sc)["R"]
    rz.async_status_reset()
    with rz.static_capacity(int(R * 1.5)):
        got = hr.run(variant, sc, og)
    st = rz.async_status(reset=True)
    assert len(st) == 1 and st[0][0] == R and st[0][1] is False and st[0][2] >= int(R * 1.5)
    for k in ("color", "radii", "others", "out_all_map", "plane_depth", "observe"):
        if k in ref:
            assert np.array_equal(ref[k], got[k]), (variant, k)          # same kernels, same order: bit-identical outputs
    for k, r in ref["grads"].items():
        if r is not None:
            assert np.linalg.norm(got["grads"][k].astype(np.float64) - r) <= 2e-5 * np.linalg.norm(r.astype(np.float64)) + 1e-30, (variant, k)
    with rz.static_capacity(max(R // 3, 1)):                                # too small on purpose
        hr.run(variant, sc, None)
    st = rz.async_status(reset=True)
    assert st[0][0] == R and st[0][1] is True


def test_iteration_replays_from_a_hip_graph():
    """forward + fused loss + backward of the surfel rasterizer recorded with torch.cuda.graph (static parameter tensors, static gradient slots)
    and replayed after the parameters moved: gradients equal those of an eager run on the moved parameters."""
    import hiprun as hr
    import diff_surfel_rasterization as dsr
    from gsrast import rasterize as rz
    from gsrast.losses import l1_plus_linear
    W, H, P = 320, 200, 8000
    sc = scenes.make_scene("surfel", P, W, H, seed=4)
    t = hr.to_dev(sc, "cuda")
    rs = hr.settings("surfel", t)
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(3, H, W, generator=g).cuda(); wmap = (torch.randn(11, H, W, generator=g) / (W * H)).cuda()
    names = ("means3D", "opacities", "colors_precomp", "scales", "rotations")
    leaves = {k: t[k].clone().requires_grad_(True) for k in names}
    m2 = torch.zeros(P, 3, device="cuda", requires_grad=True)
    one = torch.ones((), device="cuda")

    def iteration():
        color, radii, allmap = dsr.GaussianRasterizer(rs)(means2D=m2, **leaves)
        loss = l1_plus_linear(color, gt, allmap, wmap)
        loss.backward(gradient=one)
        return loss

    def grads():
        return {k: v.grad.detach().clone() for k, v in leaves.items()}

    def zero():
        for v in list(leaves.values()) + [m2]:
            v.grad = None

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):                       # eager warm-up on the side stream (also seeds the capacity hint)
            zero(); iteration()
    torch.cuda.current_stream().wait_stream(s)
    zero()
    rz.async_status_reset()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss_static = iteration()
    static_grads = {k: v.grad for k, v in leaves.items()}            # the graph writes the gradients into these tensors on every replay
    with torch.no_grad():                         # move the scene: the graph must read the NEW values from the same tensors
        leaves["means3D"].add_(0.01 * torch.randn(P, 3, generator=g).cuda())
        leaves["opacities"].mul_(0.9)
    graph.replay()
    torch.cuda.synchronize()
    got = {k: v.detach().clone() for k, v in static_grads.items()}
    got_loss = float(loss_static)
    st = rz.async_status()
    assert len(st) == 1 and st[0][1] is False and st[0][0] > 0
    zero()
    ref_loss = float(iteration())
    ref = grads()
    # the loss is a float-atomic sum over ~2000 blocks of terms that largely cancel (signed weights): its last digits depend on the order of arrival
    assert abs(got_loss - ref_loss) <= 1e-4 * abs(ref_loss) + 1e-7
    for k in names:
        d = (got[k] - ref[k]).norm().item(); n = ref[k].norm().item()
        assert d <= 2e-5 * n + 1e-30, (k, d, n)


def test_decode_static_rows_parks_the_tail_and_keeps_gradients():
    """static_rows=True: same first P rows / same gradients as the exact path, the remaining Nv*k - P rows parked at the camera centre with zero
    opacity, count = P, and a surfel rasterizer fed all rows renders the same image (the parked rows are culled: radii 0)."""
    import decode_cases
    import hiprun as hr
    from gsrast import decode
    case = decode_cases.make_case(Na=3000, seed=2)
    dev = "cuda"
    t = lambda a: None if a is None else torch.tensor(a, device=dev)
    par = {n: t(v) for n, v in case["params"].items()}
    heads = ((par["W1o"], par["b1o"], par["W2o"], par["b2o"]), (par["W1c"], par["b1c"], par["W2c"], par["b2c"]), (par["W1k"], par["b1k"], par["W2k"], par["b2k"]))
    vis = torch.tensor(case["vis_idx"], dtype=torch.int32, device=dev)
    campos = t(case["campos"])

    def run(static):
        leaves = [t(case[k]).requires_grad_(True) for k in ("anchor", "feat", "offset", "scaling")]
        out = decode.neural_gaussians(leaves[0], leaves[1], leaves[2], leaves[3], heads[0], heads[1], heads[2], campos, vis_idx=vis,
                                      appearance=par["app"], static_rows=static)
        P = int(out[7][0]) if static else out[0].shape[0]
        g = torch.Generator().manual_seed(0)
        w = [torch.randn(P, c, generator=g).to(dev) for c in (3, 3, 1, 3, 4)]
        loss = sum((o[:P] * ww).sum() for o, ww in zip(out[:5], w))
        loss.backward()
        return out, P, [l.grad.clone() for l in leaves]

    oe, Pe, ge = run(False)
    os_, Ps, gs = run(True)
    k = case["offset"].shape[1]
    assert Ps == Pe and os_[0].shape[0] == vis.numel() * k
    for a, b in zip(oe[:5], os_[:5]):
        assert torch.equal(a, b[:Pe])
    assert torch.equal(os_[0][Pe:], campos.expand(os_[0].shape[0] - Pe, 3)) and not bool(os_[2][Pe:].any())
    for a, b in zip(ge, gs):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_scaffold_iteration_replayed_from_a_graph_tracks_the_eager_one():
    """The complete scaffold-2dgs iteration (prefilter, decode, surfel rasterizer, L1+SSIM + normal / distortion + scaling losses, backward,
    densification statistics, fused Adam) recorded once with gsrast.graphs.GraphedStep and replayed: parameters and statistics after the same
    number of iterations agree with the eager, reference-shaped iteration (differences: float-atomic order only)."""
    import os
    import sys
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_pipeline
    from gsrast.graphs import GraphedStep
    dev = torch.device("cuda:0")
    n_warm, n_run = 3, 4
    e_step, e_st = bench_pipeline.build(types.SimpleNamespace(decode="hip", loss="full-hip", Na=9000), dev)
    for _ in range(n_warm + n_run):
        e_step()
    g_step, g_st = bench_pipeline.build(types.SimpleNamespace(decode="hip", loss="full-hip", Na=9000, static=True), dev)
    it = GraphedStep(g_step, optimizers=g_st["optimizers"], warmup=n_warm)
    for _ in range(n_run):
        it()
    st = it.check()
    assert len(st) == 1 and st[0][1] is False and st[0][0] > 0
    pe = [p for g in e_st["optimizers"][0].param_groups for p in g["params"]]
    pg = [p for g in g_st["optimizers"][0].param_groups for p in g["params"]]
    assert e_st["P"] == g_st["P"] and g_st["rows"] >= g_st["P"]
    for a, b in zip(pe, pg):
        assert float(e_st["optimizers"][0].state[a]["step"]) == float(g_st["optimizers"][0].state[b]["step"]) == n_warm + n_run
        d = (a.detach() - b.detach()).norm().item(); n = a.detach().norm().item()
        assert d <= 2e-3 * n + 1e-6, (tuple(a.shape), d, n)


def test_forward_async_reports_a_prefiltered_violation_and_profile_enable_values():
    """(1) the sync-free forward used to ignore `prefiltered=True` violations that the synchronous forwards fail on (the reference traps the
    device, auxiliary.h:156-160); it now sets a third sticky status word that async_status() raises on.  (2) gsr_profile_enable: any non-zero value
    without stage bits (2, -1 & 0xFF) keeps its old meaning "every stage" instead of silently switching the profiler off."""
    import hiprun as hr
    import gsrast
    from gsrast import rasterize as rz
    W, H, P = 160, 112, 1500
    sc = scenes.make_scene("surfel", P, W, H, seed=5)
    t = hr.to_dev(sc, "cuda")
    vid = hr.VID["surfel"]
    rs = hr.settings("surfel", t)._replace(prefiltered=True)
    args = lambda m: (m, None, t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], None, None)
    rz.async_status_reset()
    with rz.static_capacity(200000):
        rz.forward(vid, *args(t["means3D"]), rs)
    assert rz.async_status(reset=True)[0][1] is False                 # the scene keeps the promise: nothing raised
    bad = t["means3D"].clone()
    V = t["viewmatrix"]
    bad[7] = (torch.tensor([0.0, 0.0, -1.0], device=bad.device) - V[3, :3]) @ torch.linalg.inv(V[:3, :3])      # one gaussian behind the camera
    with rz.static_capacity(200000):
        rz.forward(vid, *args(bad), rs)
    with pytest.raises(RuntimeError, match="prefiltered is set"):
        rz.async_status(reset=True)
    with rz.static_capacity(200000):
        rz.forward(vid, *args(bad), rs._replace(prefiltered=False))   # without the promise the point is simply culled
    rz.async_status(reset=True)
    # profiler enable values
    L = gsrast.lib()
    for val in (2, 255, 1):
        L.gsr_profile_enable(val)
        rz.forward(vid, *args(t["means3D"]), rs._replace(prefiltered=False))
        torch.cuda.synchronize()
        rec = gsrast.profile_read()
        assert rec["preprocess"][1] >= 1 and rec["blend_fwd"][1] >= 1, (val, rec)
    L.gsr_profile_enable(0)


def test_overflow_of_any_replay_stays_flagged_until_checked():
    """A rasterizer forward recorded into a HIP graph keeps its status words OUTSIDE the graph (a zeroed row of a per-device pool, never cleared by
    the library): a replay that overflows the recorded binning capacity is still flagged after later replays that fit, until the owner looks."""
    import hiprun as hr
    import diff_gaussian_rasterization as dgr
    from gsrast import rasterize as rz
    W, H, P = 320, 200, 6000
    sc = scenes.make_scene("ewa", P, W, H, seed=12)
    t = hr.to_dev(sc, "cuda")
    rs = hr.settings("ewa", t)
    leaves = {k: t[k].clone() for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations")}
    m2 = torch.zeros(P, 3, device="cuda")

    def fwd():
        with torch.no_grad():
            return dgr.GaussianRasterizer(rs)(means2D=m2, **leaves)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            ref_color = fwd()[0].clone()         # eager: seeds the capacity hint and the status pool
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    rz.async_status_reset()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fwd()
    rows = list(rz._ASYNC_STATUS)
    rz.async_status_reset()
    assert len(rows) == 1
    graph.replay(); torch.cuda.synchronize()
    h = rows[0][0].cpu()
    R = int(h[0])
    assert R > 0 and int(h[1]) == 0 and torch.equal(out[0], ref_color)
    keep = leaves["scales"].clone()
    leaves["scales"].mul_(5.0)                    # ~25 x the footprint: far beyond the recorded capacity
    graph.replay()
    leaves["scales"].copy_(keep)
    graph.replay(); torch.cuda.synchronize()
    h = rows[0][0].cpu()
    assert int(h[0]) == R and int(h[1]) == 1      # the last replay fitted; the overflow of the one before is still on record
    assert torch.equal(out[0], ref_color)
    rows[0][0].zero_()
    graph.replay(); torch.cuda.synchronize()
    assert int(rows[0][0].cpu()[1]) == 0
