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
    assert abs(got_loss - ref_loss) <= 1e-5 * abs(ref_loss) + 1e-7
    for k in names:
        d = (got[k] - ref[k]).norm().item(); n = ref[k].norm().item()
        assert d <= 2e-5 * n + 1e-30, (k, d, n)
