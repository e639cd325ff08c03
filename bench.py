#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: train iters/sec + rasterize fwd+bwd ms @ 300k Gaussians, 1920x1080.

A "step" is one training iteration of the hot path on one synthetic tile-scene resident in HBM:
    drop-in rasterizer forward -> image-space loss -> autograd backward (rasterizer backward) -> fused Adam step.
Default workload = BASELINE.json configs[1]: scaffold-2dgs => diff_surfel_rasterization with colors_precomp
(SURVEY.md §8 table), P = 300 000, 1920x1080, synthetic scene of SURVEY.md §8d.  The loss touches every auxiliary
channel the 2DGS scene uses (rgb L1, alpha, depth, normal, distortion: gssr/scene/twodgs_scene.py:25-35,88-105), so
every gradient path of the backward kernel is live.  Nothing is skipped inside the timed region.

Multi-GPU: VastGaussian tiles are independent sub-scenes -> one tile per GPU, one process per GPU, no
data-path collective ("weak" scaling); value = all ranks' iterations / max-over-ranks time.

How N > 1 is launched.  Either form gives N ranks, one per GPU, RCCL (backend "nccl") for the barrier / reductions:
  * `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
    (RANK / LOCAL_RANK / WORLD_SIZE in the environment: this process IS one rank);
  * `python bench.py --gpus N ...` with no RANK in the environment: this process becomes the launcher and starts the N ranks itself
    through gsrast.launch_tiles.spawn_ranks (HIP_VISIBLE_DEVICES + NUMA affinity per child, children polled, first failure ends the job).
    N > torch.cuda.device_count() is an error unless --oversubscribe is given; ranks then share devices and, because RCCL refuses two
    ranks on one device, the control collectives run over gloo on host tensors (reported as "backend" in the line).

One JSON line on rank 0, with "roofline" (dominant kernel = blend backward, live HIP-event duration from the library's
stage profiler, algorithmic bytes of SURVEY.md §8d) and "cpu_baseline" (the CPU oracle, kind "port", N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "gs-sr_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes(variant, R, N, T):
    """blend fwd / bwd algorithmic bytes per launch."""
    rec = {"ewa": 40, "plane": 60, "surfel": 76}[variant]
    pix_f = {"ewa": 20, "plane": 44, "surfel": 76}[variant]
    grad = {"ewa": 44, "plane": 68, "surfel": 72}[variant]
    pix_b = {"ewa": 20, "plane": 64, "surfel": 76}[variant]
    fwd = R * rec + N * pix_f + 8 * T
    bwd = R * rec + R * grad + N * pix_b + 8 * T
    return fwd, bwd


def depth_order_is_global(P, T, variant="surfel"):
    """The library's static rule (gsr_binning.hip gsr_depth_order_static_rule): per-tile depth sort while P <= 192 T (surfel; 176 T plane, 155 T ewa) unless GSR_DEPTH_ORDER says otherwise.
    (On scenes with tile lists beyond 6000 entries the library's feedback switches to the global order, gsr_api.hip gsr_forward_begin: the skewed side
    scenes of --skew-frac; not the BASELINE workload.)"""
    e = os.environ.get("GSR_DEPTH_ORDER", "")
    return True if e[:1] == "g" else (False if e[:1] == "t" else P > {"ewa": 155, "plane": 176}.get(variant, 192) * T)


def stage_bytes(variant, color_mode, P, R, N, T):
    """Algorithmic bytes per launch of every stage (sort passes as implemented here: ceil(log2 T / 8) tile passes over R;
    depth order either 4 radix passes over P (key+value, read+write) or, per-tile mode, one pass over R: id + depth-key gather read, id written)."""
    sh = 192 if color_mode == "sh" else 12
    fwd_b, bwd_b = algorithmic_bytes(variant, R, N, T)
    pre = P * ((12 + 8 + 16 + 4 + sh) + (4 + 8 + 4 + 36 + 16 + 4)) if variant == "surfel" else \
        P * ((12 + 12 + 16 + 4 + sh) + (4 + 8 + 4 + 24 + 16 + 12 + 4 + 3))
    tile_bits = max(1, (T - 1).bit_length())
    tile_passes = (tile_bits + 7) // 8
    acc = {"ewa": 48, "plane": 64, "surfel": 80}[variant]
    pre_bwd = P * (acc + 12 + 16 + 8 + (2 * 192 if color_mode == "sh" else 0) + 12 + 12 + 12 + 4 + 36 + 8 + 16)
    glob = depth_order_is_global(P, T, variant)
    tile_sort = 0 if glob else 12 * R          # per-tile depth sort: ids read + depth keys gathered + ids written
    fused = os.environ.get("GSR_TILE_SORT", "fused")[:1] != "k"      # ... in k_blend_fwd's prologue (default) or as its own launch in the binning stage
    return {"preprocess": pre, "depth_order": (4 * 16 * P + 8 * P) if glob else 8 * P,
            "binning": 20 * P + 8 * R + tile_passes * 16 * R + 4 * R + 8 * T + (0 if fused else tile_sort),
            "blend_fwd": fwd_b + (tile_sort if fused else 0), "bwd_memset": P * acc, "blend_bwd": bwd_b, "preprocess_bwd": pre_bwd}


def clock_prewarm(device, ms):
    """Keep every CU busy for `ms` milliseconds with work that is NOT the measured path (torch.polygamma over 16 M floats: ~100 VALU
    instructions per element), so that the shader clock has left its idle state before the W warm-up steps start.  Measured
    (tools/step_series.py, profiles/r03_clock_ramp.txt): from an idle device the VALU-bound blend kernels need ~20 iterations (~20 ms of
    load) to come down from 0.53 to 0.475 ms while the HBM-bound kernels of the same iterations (loss, Adam, radix scatter) do not move at
    all -- the shader clock ramps, not the workload -- so `--steps 20 --warmup 5` from idle reads ~4.5 % below `--steps 500`.  A light or
    launch-bound load (sin over 4 M floats, a 1 GiB scale) does not lift the clock; 200 ms of this one removes most of the ramp.
    `--clock-prewarm-ms 0` measures from idle.  Returns the milliseconds spent."""
    if ms <= 0:
        return 0.0
    x = torch.rand(1 << 24, device=device) + 1.0
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(16):
            torch.polygamma(2, x)
        torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) * 1e3


def make_step(variant, sc, device):
    """One training iteration.  All gaussian parameters live in ONE flat leaf z (contiguous blocks: means 3P, scales 2P|3P,
    rotations 4P, opacity P, colour 3P|48P), optimised by gsrast.optim.Adam (one fused HIP kernel, include/gsrast.h gsr_adam_step) whose
    per-element `lr_scale` carries the per-group rates of gssr/gaussian/*.setup_optimizers -- one optimizer kernel instead of one per group.
    (GSR_BENCH_TORCH_ADAM=1 selects the earlier form: params = z * lr_scale under torch's fused Adam with lr = 1.)
    The auxiliary-map loss is linear in the 11 (5) channels so autograd hands the rasterizer a dense dL_dothers without
    materialising one zero-padded [11,H,W] tensor per sliced channel; every gradient path of the backward kernel is live."""
    from gsrast import runner as hiprun
    import diff_gaussian_rasterization as dgr
    import diff_surfel_rasterization as dsr
    import diff_plane_rasterization as dpr
    from gsrast.losses import l1_plus_linear
    t = hiprun.to_dev(sc, device)
    rs = hiprun.settings(variant, t)
    P, W, H = t["means3D"].shape[0], int(t["W"]), int(t["H"])
    ns = t["scales"].shape[1]
    use_sh = t.get("shs") is not None
    # Learning rates: the reference's per-group ratios, scaled so that 100+ Adam steps against the random target (or 2000) do not move the scene away
    # from the SURVEY 8d distribution the workload is defined on (Adam's step is ~lr per iteration whatever the gradient: with the
    # reference's absolute rates on these post-activation parameters the splat sizes random-walk by +-50 % within 100 steps, and the tile
    # instance count -- hence the work per step -- with them).  The optimiser work per step is the same; config reports R before / after.
    cols = [("means3D", 3, 1.6e-8), ("scales", ns, 5e-7), ("rotations", 4, 1e-7), ("opacities", 1, 1e-6)]
    cols.append(("shs", 48, 2.5e-6) if use_sh else ("colors_precomp", 3, 2.5e-6))
    # One leaf per parameter tensor, as the reference's models hold them (vanilla_gaussian.py:120-139), every one with its own learning rate;
    # gsrast.optim.Adam updates all of them in ONE launch (gsr_adam_step_multi).  (Rounds 1-2 kept one flat leaf + torch.split, whose backward is a
    # cat of the five gradients: 7 us + a launch gap per iteration; GSR_BENCH_FLAT=1 selects that form, GSR_BENCH_TORCH_ADAM=1 torch's fused Adam on it.)
    sizes = [P * n for _, n, _ in cols]
    torch_adam = os.environ.get("GSR_BENCH_TORCH_ADAM", "0") == "1"
    flat = torch_adam or os.environ.get("GSR_BENCH_FLAT", "0") == "1"
    leaves = None
    if torch_adam:
        lr_scale = torch.cat([torch.full((P * n,), lr, device=device) for _, n, lr in cols])
        z = (torch.cat([t[k].reshape(-1) for k, _, _ in cols]) / lr_scale).clone().requires_grad_(True)
        opt = torch.optim.Adam([z], lr=1.0, eps=1e-15, fused=True)
    elif flat:
        from gsrast.optim import Adam
        lr_scale = torch.cat([torch.full((P * n,), lr, device=device) for _, n, lr in cols])
        z = torch.cat([t[k].reshape(-1) for k, _, _ in cols]).clone().requires_grad_(True)
        opt = Adam([{"params": [z], "lr": 1.0, "lr_scale": lr_scale}], lr=0.0, eps=1e-15)
    else:
        from gsrast.optim import Adam
        leaves = {k: t[k].reshape(P, n).clone().requires_grad_(True) for k, n, _ in cols}
        opt = Adam([{"params": [leaves[k]], "lr": lr, "name": k} for k, _, lr in cols], lr=0.0, eps=1e-15)
    g = torch.Generator(device="cpu").manual_seed(1234)
    gt = torch.rand((3, H, W), generator=g).to(device)
    N = float(W * H)
    gtn = torch.nn.functional.normalize(torch.randn((3, H, W), generator=g), dim=0)
    if variant == "surfel":
        wmap = torch.zeros((11, H, W))
        wmap[0] = 0.01 / N; wmap[1] = 0.01 / N; wmap[2:5] = -0.05 * gtn / N; wmap[5] = 0.01 / N; wmap[6] = 100.0 / N
        wmap[8:11] = 0.0        # median-normal channels: left without upstream gradient, as in twodgs_scene.py:88-105
        wmap = wmap.to(device)
    elif variant == "plane":
        wmap = torch.zeros((5, H, W)); wmap[0:3] = -0.05 * gtn / N; wmap[3] = 0.01 / N; wmap[4] = 0.01 / N
        wmap = wmap.to(device)
        wpd = torch.full((1, H, W), 0.01 / N, device=device)
    all_map = t.get("all_map")
    state = {}
    # screen-space gradient carriers: the rasterizer only uses their .grad slot, so the leaves persist across iterations
    means2D = torch.zeros((P, 3), dtype=torch.float32, device=device, requires_grad=True)
    m2a = torch.zeros((P, 3), dtype=torch.float32, device=device, requires_grad=True) if variant == "plane" else None

    one = torch.ones((), dtype=torch.float32, device=device)          # d(loss)/d(loss): what loss.backward() would otherwise fill per call

    def step():
        if leaves is not None:
            v = leaves
        else:
            prm = z * lr_scale if torch_adam else z
            parts = torch.split(prm, sizes)                                 # backward = one cat, not one zero-pad per slice
            v = {k: parts[i].view(P, n) for i, (k, n, _) in enumerate(cols)}
        kw = dict(means3D=v["means3D"], means2D=means2D, opacities=v["opacities"], scales=v["scales"], rotations=v["rotations"])
        if use_sh:
            kw["shs"] = v["shs"].reshape(P, 16, 3)
        else:
            kw["colors_precomp"] = v["colors_precomp"]
        if variant == "surfel":
            color, radii, allmap = dsr.GaussianRasterizer(rs)(**kw)
            loss = l1_plus_linear(color, gt, allmap, wmap, root=True)
        elif variant == "plane":
            color, radii, observe, oam, pd = dpr.GaussianRasterizer(rs)(means2D_abs=m2a, all_map=all_map, **kw)
            loss = l1_plus_linear(color, gt, oam, wmap, root=True) + (pd * wpd).sum()
        else:
            color, radii = dgr.GaussianRasterizer(rs)(**kw)
            loss = l1_plus_linear(color, gt, root=True)
        loss.backward(gradient=one)
        opt.step()
        opt.zero_grad(set_to_none=True)
        state["viewspace_grad"] = means2D.grad          # what densification reads (viewspace_points.grad[:, :2])
        means2D.grad = None
        if m2a is not None:
            m2a.grad = None
        state["vis"] = radii
        return loss

    def current_scene():
        """The scene dict with the parameters as they are now (for the R-after figure)."""
        cur = dict(sc)
        with torch.no_grad():
            parts = [leaves[k] for k, _, _ in cols] if leaves is not None else torch.split(z * lr_scale if torch_adam else z, sizes)
            for i, (k, n, _) in enumerate(cols):
                a = parts[i].view(P, n).cpu().numpy()
                cur[k] = a.reshape(P, 16, 3) if k == "shs" else (a.reshape(P) if k == "opacities" and sc[k].ndim == 1 else a)
        return cur
    state["current_scene"] = current_scene
    state["optimizer"] = opt
    return step, state


def cpu_baseline(variant, sc, og, budget_s=25.0):
    """Times the CPU oracle (plain-C restatement of the reference's rasterizer, OpenMP) on the SAME scene, as BASELINE.md section 3 asks:
      * all host threads: rasterizer forward+backward (`value`), and one complete TRAIN STEP = that + the L1 image loss / its gradient + Adam on
        the explicit parameters, the latter two as torch CPU ops (`train_step_ms`);
      * one thread: a bounded sample -- every 32nd tile in the blend loops -- extrapolated to the whole image (`one_thread`).
    Also returns workload statistics of the forward (evaluated / contributing (pixel, splat) pairs)."""
    import oracle
    keep = {}

    def once():
        with oracle.Forward(sc, variant) as f:
            f.backward(**og)
            keep["pairs"] = f.pair_counts()
    t0 = time.time()
    once()
    first = time.time() - t0
    n, tot = 0, 0.0
    while tot < budget_s - first and n < 5:
        t0 = time.time()
        once()
        tot += time.time() - t0
        n += 1
    per = (tot / n) if n else first
    full_pairs = keep["pairs"]                 # of a full run (the one-thread sample below overwrites keep)
    # loss + optimizer of the train step on the host (torch CPU, all threads): L1 against a target + its gradient, Adam on every parameter tensor
    P, W, H = sc["means3D"].shape[0], int(sc["W"]), int(sc["H"])
    img = torch.rand(3, H, W); gt = torch.rand(3, H, W)
    prm = [torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(sc[k], np.float32)).reshape(P, -1).clone())
           for k in ("means3D", "scales", "rotations", "opacities", "shs" if sc.get("shs") is not None else "colors_precomp")]
    opt = torch.optim.Adam(prm, lr=1e-4, eps=1e-15)
    for q in prm:
        q.grad = torch.randn_like(q)
    t0 = time.time()
    for _ in range(3):
        x = img.clone().requires_grad_(True)
        (x - gt).abs().mean().backward()
        opt.step()
    extra = (time.time() - t0) / 3
    # one thread, every 32nd tile; the per-Gaussian stages (preprocess, scan, duplicate, sort) are measured with no tile at all and not scaled
    T = ((W + 15) // 16) * ((H + 15) // 16)
    threads = oracle.omp_threads()
    oracle.set_threads(1)
    try:
        oracle.set_tile_stride(T + 1); t0 = time.time(); once(); t_fixed = time.time() - t0
        oracle.set_tile_stride(32); t0 = time.time(); once(); t_s = time.time() - t0
    finally:
        oracle.set_tile_stride(1); oracle.set_threads(threads)
    one = t_fixed + 32.0 * max(t_s - t_fixed, 0.0)
    return {"value": round(1.0 / per, 4), "unit": "rasterize fwd+bwd iters/s", "cores": threads, "kind": "port",
            "ms_per_iter": round(per * 1e3, 1),
            "train_step_ms": round((per + extra) * 1e3, 1), "train_step_iters_per_s": round(1.0 / (per + extra), 4),
            "train_step_what": f"rasterizer fwd+bwd ({per * 1e3:.0f} ms, oracle, {threads} threads) + L1 loss and gradient + Adam on {sum(q.numel() for q in prm)} "
                               f"parameters ({extra * 1e3:.1f} ms, torch CPU)",
            "one_thread": {"ms_per_iter_extrapolated": round(one * 1e3, 0), "iters_per_s": round(1.0 / one, 5),
                           "sample": f"1 OpenMP thread; per-Gaussian stages in full ({t_fixed:.2f} s) + every 32nd tile of the blend forward+backward "
                                     f"({t_s - t_fixed:.2f} s) x 32"},
            "sample": f"{max(n, 1)} x full workload ({variant}, same scene as the GPU run), oracle/gsr_oracle.c with OpenMP; "
                      f"rasterizer forward+backward only (no loss/optimizer) for `value`"}, {"pairs": full_pairs}


def parity_full_size(variant, sc, og, device, color_mode):
    """HIP rasterizer on the BASELINE workload itself (same scene, same upstream gradients) against the FLOAT64 truth
    (oracle/libgsr_oracle_f64.so on the integer stages of the float32 oracle), with the float32 oracle's own error against the same truth
    beside every figure -- the criterion of tests/test_gpu_parity.py::test_full_size_oracle_parity (tests/parity_truth.py), reported here."""
    from gsrast import runner as hiprun
    import parity_truth as pt
    st = hiprun.run_raw(variant, sc, device=device)
    res = hiprun.run(variant, sc, og, device=device)
    lists = "pass"
    try:
        # the library's tile-instance list (instances that can reach no pixel of their tile are not emitted, gsr_tile_cull.h) against the oracle's full
        # list: subset in order, nothing contributing dropped (float32 and float64), the exact float64 region kept -- tests/tile_cull.py
        f32, fma, truth, ints = pt.run_oracles(sc, variant, og, hip_state=st)
    except AssertionError as e:
        lists = "FAIL: " + str(e)[:300]
        f32, fma, truth, ints = pt.run_oracles(sc, variant, og)
    view = ints.get("view")
    cand = dict(color=st["color"], final_T=st["final_T"], n_contrib=view["n_contrib"] if view else st["n_contrib"], grads=res["grads"])
    if variant == "surfel":
        cand["others"] = st["others"]
    if variant == "plane":
        cand.update(all_map=st["all_map"], plane_depth=st["plane_depth"], observe=st["observe"])
    rep, verdict = {}, "pass"
    try:
        pt.check_case(variant, color_mode, cand, f32, fma, truth, rep)
    except AssertionError as e:
        verdict = "FAIL: " + str(e)[:300]
    rnd = lambda v: float(f"{v:.4g}") if isinstance(v, float) else v
    out = {"criterion": "tests/parity_truth.py vs float64 truth", "verdict": verdict,
           "radii_equal": bool(np.array_equal(st["radii"], ints["radii"])), "instance_list_vs_oracle": lists,
           "tile_instances": {k: (rnd(v) if isinstance(v, float) else v) for k, v in (view or {}).items() if k not in ("keep", "n_contrib")},
           "robust_pixel_fraction": rnd(rep.get("robust_pixel_fraction", 0.0)), "fragile_pixels_by_gate": rep.get("fragile_pixels_by_gate"),
           "robust_row_fraction": rnd(rep.get("robust_row_fraction", 0.0))}
    for k in ("n_contrib", "median_contributor", "median_splat", "observe"):
        if k in rep:
            out[k] = rep[k]
    out["maps_err_vs_f64"] = {k: {"hip_robust_max": rnd(v["robust_max"]), "f32_oracle_robust_max": rnd(v["oracle_robust_max"]),
                                  "hip_robust_px_beyond_tol": v["robust_px_beyond_tol"], "f32_oracle_robust_px_beyond_tol": v["oracle_robust_px_beyond_tol"],
                                  "hip_fragile_px_beyond_tol": v["fragile_px_beyond_tol"], "f32_oracle_fragile_px_beyond_tol": v["oracle_fragile_px_beyond_tol"],
                                  "f32_geometry_floor_robust_px_beyond_tol": v.get("floor_robust_px_beyond_tol"), "f32_geometry_floor_robust_max": rnd(v.get("floor_robust_max", 0.0)),
                                  "hip_robust_px_beyond_tol_vs_floor_run": v.get("robust_px_beyond_tol_vs_floor"), "hip_robust_max_vs_floor_run": rnd(v.get("robust_max_vs_floor", 0.0))}
                              for k, v in rep.items() if isinstance(v, dict) and "robust_max" in v and k in ("color", "final_T", "others[0]", "others[2]", "others[6]", "all_map", "plane_depth")}
    out["grad_rel_l2_vs_f64"] = {k: {"hip": rnd(v["rel_l2"]), "f32_oracle": rnd(v["oracle_rel_l2"]), "hip_all_rows": rnd(v["rel_l2_all_rows"]),
                                     "f32_oracle_all_rows": rnd(v["oracle_rel_l2_all_rows"]),
                                     # the float64 blend of the FLOAT32 per-gaussian state (tests/parity_truth.py "THE FLOOR"): what the reference's own float32
                                     # preprocess costs any blend implementation, and how far the HIP blend is from exact arithmetic on that state
                                     "f32_geometry_floor": rnd(v.get("floor_rel_l2", 0.0)), "hip_vs_floor_run": rnd(v.get("rel_l2_vs_floor", 0.0)),
                                     "f32_oracle_vs_floor_run": rnd(v.get("oracle_rel_l2_vs_floor", 0.0))}
                                 for k, v in rep.items() if isinstance(v, dict) and "rel_l2" in v}
    return out


def measured_copy_gbs(device, nbytes=1 << 30):
    """Device-to-device float4 copy bandwidth (read + write bytes / time): the achievable-HBM figure beside the 8 TB/s spec."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize(device)
    return 2.0 * nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def method_iteration(device, which, steps=20):
    """Extra, informational: the COMPLETE training iteration of a BASELINE config around the same rasterizer, every op a HIP kernel of
    this repo, with GPU-kernel time (torch.profiler / roctracer: every kernel of the process, C-ABI launches included) beside wall time.
      scaffold-2dgs (configs[1]): prefilter (scaffold_filter) -> neural-Gaussian decode (72k anchors x 10 offsets -> ~320k Gaussians) ->
        diff_surfel_rasterization -> L1+SSIM, normal + distortion regularisers, scaling loss -> backward -> densification statistics -> Adam.
      octree-2dgs (configs[3] / [4]): the same iteration behind the Octree model's level-of-detail mask + prefilter (87k anchors on 6 levels).
      octree-pgsr (configs[2], after step 7000), for the view AND its neighbour camera: octree level-of-detail mask + prefilter ->
        neural-Gaussian decode (74k anchors x 10 offsets on 6 levels -> ~300k Gaussians) -> per-Gaussian all_map -> diff_plane_rasterization;
        then L1+SSIM + single-view normal loss + multi-view geometric / NCC losses + scaling loss -> backward -> statistics -> Adam.
      pgsr: the same losses and two renders on P = 300k explicit Gaussians (vanilla PGSR, no decode).
    `value` above times the hot path itself (rasterizer + image loss + Adam on explicit Gaussians), which the roofline / stage figures refer to."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import iter_breakdown
    if which in ("scaffold-2dgs", "octree-2dgs"):
        import bench_pipeline
        lod = which == "octree-2dgs"
        step, st = bench_pipeline.build(types.SimpleNamespace(decode="hip", loss="full-hip", Na=87000 if lod else 72000, lod=lod), device)
    elif which == "octree-pgsr":
        import bench_pipeline_octree_pgsr
        step, st = bench_pipeline_octree_pgsr.build(types.SimpleNamespace(Na=74000), device)
    else:
        import bench_pipeline_pgsr
        step, st = bench_pipeline_pgsr.build(types.SimpleNamespace(glue="hip", P=300000), device)
    try:
        r = iter_breakdown.measure(step, device, steps=steps, warmup=25, top=8)
    except Exception as e:       # profiler unavailable: wall time only
        for _ in range(8):
            step()
        torch.cuda.synchronize(device); t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(device)
        r = {"wall_ms": round(1e3 * (time.perf_counter() - t0) / steps, 4), "kernel_ms": None, "profiler_error": str(e)[:80]}
    r.update(method=which, gaussians=int(st["P"]), iters_per_s=round(1e3 / r["wall_ms"], 2), mode="eager, reference-shaped (exact row counts, one host sync per decode)")
    del step, st
    # the same iteration in its static-shape form, recorded once into a HIP graph and replayed (gsrast.graphs.GraphedStep)
    try:
        from gsrast.graphs import GraphedStep
        if which in ("scaffold-2dgs", "octree-2dgs"):
            gstep, gst = bench_pipeline.build(types.SimpleNamespace(decode="hip", loss="full-hip", Na=87000 if lod else 72000, lod=lod, static=True), device)
        elif which == "octree-pgsr":
            gstep, gst = bench_pipeline_octree_pgsr.build(types.SimpleNamespace(Na=74000, static=True), device)
        else:
            gstep, gst = bench_pipeline_pgsr.build(types.SimpleNamespace(glue="hip", P=300000), device)
        it = GraphedStep(gstep, optimizers=gst["optimizers"], warmup=5)
        for _ in range(5):
            it()
        torch.cuda.synchronize(device); t0 = time.perf_counter()
        for _ in range(3 * steps):
            it()
        torch.cuda.synchronize(device)
        wall = 1e3 * (time.perf_counter() - t0) / (3 * steps)
        r["graph_replay"] = {"wall_ms": round(wall, 4), "iters_per_s": round(1e3 / wall, 2), "rasterizer_forwards": [list(x) for x in it.check()],
                             "what": "static-shape iteration (decode static_rows, sync-free rasterizer forward, Adam scalars from device memory) replayed from one HIP graph"}
    except Exception as e:
        r["graph_replay"] = {"error": str(e)[:200]}
    return r


def child_env():
    """Environment of the single-process helper children: no torchrun variables (they must not join or shadow the parent's process group)."""
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                             "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}


def launch_ranks(args, ndev):
    """`python bench.py --gpus N` with no RANK in the environment: start the N ranks (the job train_split.py:24-38 runs one tile after the
    other, and train.py:78-80 refuses on more than one GPU) and return the job's exit code; rank 0 prints the JSON line."""
    from gsrast import launch_tiles
    if getattr(args, "dry_run", False):
        ndev = args.gpus                    # no device is touched: pretend one per rank, so that the pinning of an N-GPU node is what gets exercised
    if args.gpus > ndev and not args.oversubscribe:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} HIP device(s) visible; pass --oversubscribe to let ranks share devices")
    shared = args.gpus > ndev
    env = {"GSR_BENCH_BACKEND": "gloo" if (shared or getattr(args, "dry_run", False)) else "nccl"}
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    port = int(os.environ.get("MASTER_PORT", "0")) or launch_tiles.free_port()
    return launch_tiles.spawn_ranks(lambda r: cmd, args.gpus, min(args.gpus, ndev), port, pin_gpus=not args.no_pin, extra_env=env)


def dry_run(args):
    """`--dry-run`: everything of a multi-rank run EXCEPT the device -- the environment contract (RANK / WORLD_SIZE / MASTER_*), the per-rank
    HIP_VISIBLE_DEVICES pinning, the process group, the all-gathered rank identities, W untimed + K timed 'steps' (a 1 ms sleep) between barriers, the
    max-over-ranks / sum-over-ranks reduction and ONE JSON line from rank 0.  The line cannot be mistaken for a measurement: value null, metric 'DRY RUN'."""
    import torch.distributed as dist
    from gsrast import tiles
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N (see --help)")
    ranks_seen = None
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, {"rank": rank, "pid": os.getpid(), "visible": os.environ.get("HIP_VISIBLE_DEVICES"), "local_rank": os.environ.get("LOCAL_RANK")})
    for _ in range(args.warmup):
        time.sleep(0.001)
    tiles.barrier(None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    tiles.barrier(None)
    elapsed, total = tiles.reduce_job(time.perf_counter() - t0, args.steps, None)
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN -- no device work, not a measurement", "value": None, "unit": "iters/s", "dry_run": True, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "backend": "gloo", "dist_world_size": dist.get_world_size() if dist.is_initialized() else 1, "ranks": ranks_seen,
                          "total_steps_over_ranks": total, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "data": "none", "config": {"workload": "none (dry run of the launch path)", "parallelism": f"{world} ranks, 1 per GPU, no collective"}}),
              flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1,
                    help="number of ranks = GPUs (one tile-scene each).  N > 1 without RANK in the environment: bench.py starts the N ranks itself "
                         "(one pinned process per GPU, RCCL); under torch.distributed.run it is one of them")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow --gpus N > visible devices: ranks share devices (rank r on device r %% devices), control collectives over gloo")
    ap.add_argument("--dry-run", action="store_true",
                    help="NO device work, NO measurement: the N ranks are started, pinned and joined exactly as for a real run (gloo), 'step' is a 1 ms sleep, "
                         "rank 0 prints a line whose value is null and whose metric says DRY RUN.  Runs without a GPU: how the CPU test suite covers --gpus 8")
    ap.add_argument("--sync-free", action="store_true",
                    help="experiment: the timed steps run inside gsrast.rasterize.static_capacity() -- the rasterizer forward never reads num_rendered back (an arena overflow is "
                         "flagged, checked after the timed region, instead of repaired), so the host can run ahead of the device by more than one iteration")
    ap.add_argument("--no-pin", action="store_true", help="self-launch without HIP_VISIBLE_DEVICES / NUMA pinning (LOCAL_RANK selects the device)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--variant", default="surfel", choices=["ewa", "surfel", "plane"])
    ap.add_argument("--P", type=int, default=300000)
    ap.add_argument("--W", type=int, default=1920)
    ap.add_argument("--H", type=int, default=1080)
    ap.add_argument("--color-mode", default="precomp", choices=["precomp", "sh"],
                    help="precomp = scaffold/octree path (configs 2-5, default); sh = vanilla path with degree-3 SH (config 1)")
    ap.add_argument("--skew-frac", type=float, default=0.0, help="side experiment: this fraction of the gaussians is pulled towards the image centre (see concentrate())")
    ap.add_argument("--skew-scale", type=float, default=0.15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph-replay", action="store_true", help="skip the informational HIP-graph replay of the same step")
    ap.add_argument("--graph-replay-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--method-iteration-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--clock-prewarm-ms", type=float, default=300.0,
                    help="busy the device this long with unrelated compute before the warm-up steps so the shader clock is out of idle (0 = measure from idle); see clock_prewarm()")
    ap.add_argument("--stage-steps", type=int, default=20, help="untimed iterations after the timed region in which EVERY stage carries HIP events (stage_ms)")
    ap.add_argument("--profile-all-stages-in-timed-region", action="store_true",
                    help="round-1/2 behaviour: all seven stages timed with HIP events inside the timed region (costs ~6 %% of the step in event gaps)")
    ap.add_argument("--no-method-iteration", action="store_true",
                    help="skip the extra 'method_iteration' measurement (full scaffold-2dgs iteration incl. decode, real losses, statistics)")
    args = ap.parse_args()

    if args.dry_run:
        if "RANK" not in os.environ and args.gpus > 1:
            sys.exit(launch_ranks(args, 0))
        return dry_run(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product has no CPU path")
    ndev = torch.cuda.device_count()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, ndev))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N (see --help)")
    shared_devices = os.environ.get("GSR_BENCH_BACKEND", "nccl") != "nccl"
    if local_rank >= ndev and not (shared_devices or args.oversubscribe):
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {ndev} visible device(s); pass --oversubscribe to share devices")
    dev_index = local_rank % ndev                           # one GPU per rank; wraps only when devices are shared on purpose
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    backend = None
    ranks_seen = None
    if world > 1 or "RANK" in os.environ:      # one of N ranks: RCCL for the barrier / max-reduction only
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")        # "gloo": ranks share devices (RCCL refuses two ranks on one device)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # what lets a reader check that the communicator really holds N ranks on N devices: every rank's device identity, all-gathered
        pr = torch.cuda.get_device_properties(device)
        me = {"rank": rank, "pid": os.getpid(), "visible": os.environ.get("HIP_VISIBLE_DEVICES"), "device_index": dev_index,
              "uuid": str(getattr(pr, "uuid", "")), "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                                                                getattr(pr, "pci_device_id", 0))}
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)

    import gsrast
    from gsrast import workloads as scenes
    gsrast.lib()
    if args.method_iteration_child:
        print(json.dumps(method_iteration(device, args.method_iteration_child, steps=60)), flush=True)
        return
    if args.graph_replay_child:
        sc = scenes.make_scene(args.variant, args.P, args.W, args.H, seed=0, color_mode=args.color_mode)
        step, state = make_step(args.variant, sc, device)
        from gsrast.graphs import GraphedStep
        it = GraphedStep(step, optimizers=[state["optimizer"]], warmup=max(3, args.warmup))
        for _ in range(5):
            it()
        torch.cuda.synchronize(device); tg = time.perf_counter()
        for _ in range(args.steps):
            it()
        torch.cuda.synchronize(device)
        tg = time.perf_counter() - tg
        print(json.dumps({"iters_per_s": round(args.steps / tg, 3), "ms_per_step": round(1e3 * tg / args.steps, 4), "steps": args.steps,
                          "rasterizer_forwards": [list(x) for x in it.check()],
                          "what": "the step of `value` (same scene, seed 0), recorded into one HIP graph and replayed; separate process"}), flush=True)
        return
    # one independent tile-scene per rank (train_split.py trains tiles independently; seed = tile index)
    sc = scenes.make_scene(args.variant, args.P, args.W, args.H, seed=rank, color_mode=args.color_mode)
    if args.skew_frac > 0:      # side experiment (not the BASELINE workload): a fraction of the gaussians pulled towards the image centre -> a few very long tile lists
        scenes.concentrate(sc, args.skew_frac, args.skew_scale)
    step, state = make_step(args.variant, sc, device)

    from gsrast import tiles

    def barrier():
        tiles.barrier(device)

    prewarm_ms = clock_prewarm(device, args.clock_prewarm_ms)
    import contextlib
    from gsrast import rasterize as rz
    for _ in range(args.warmup):
        step()
    barrier()
    sync_free = rz.static_capacity() if args.sync_free else contextlib.nullcontext()      # (capacity: 1.25 x the running maximum of the warm-up forwards + 16384)
    rz.async_status_reset()
    # Inside the timed region only the DOMINANT kernel's stage carries HIP events (the live duration behind `roofline`): every timed stage
    # costs two event records = ~10 us of stream idle time per launch (profiles/r03_timeline.json: 82 us of gaps per 1043 us iteration with all
    # seven stages timed).  The other stages are timed in a second, untimed pass of --stage-steps iterations right after.
    gsrast.profile_enable(True, stages=None if args.profile_all_stages_in_timed_region else ["blend_bwd"])
    t0 = time.perf_counter()
    with sync_free:
        for _ in range(args.steps):
            step()
    barrier()
    elapsed = time.perf_counter() - t0
    if args.sync_free:
        st_async = rz.async_status(reset=True)
        if len(st_async) != args.steps or any(o for _, o, _ in st_async):
            raise SystemExit(f"bench.py --sync-free: {sum(1 for _, o, _ in st_async if o)} of {len(st_async)} forwards overflowed their arena")
    prof_timed = gsrast.profile_read()
    gsrast.profile_enable(False)
    reduce_dev = device if os.environ.get("GSR_BENCH_BACKEND", "nccl") == "nccl" else None
    elapsed, total_iters = tiles.reduce_job(elapsed, args.steps, reduce_dev)
    prof = prof_timed
    sync_free_info = None
    if rank == 0 and world == 1 and not args.sync_free:
        # informational: the same K steps with the rasterizer forward sync-free (gsrast.rasterize.static_capacity: num_rendered stays on the device, an arena overflow is
        # flagged instead of repaired) -- the host may then run ahead of the device by more than one iteration, so the figure does not depend on how promptly a shared
        # host reacts to the forward's mailbox word.  `value` above stays the default drop-in call (one host read of num_rendered per forward, as the reference's
        # cudaMemcpy at rasterizer_impl.cu:281).
        rz.async_status_reset()
        barrier()
        ts = time.perf_counter()
        with rz.static_capacity():
            for _ in range(args.steps):
                step()
        barrier()
        ts = time.perf_counter() - ts
        st_async = rz.async_status(reset=True)
        sync_free_info = {"iters_per_s": round(args.steps / ts, 3), "ms_per_step": round(1e3 * ts / args.steps, 4), "steps": args.steps,
                          "arena_overflows": sum(1 for _, o, _ in st_async if o),
                          "what": "the step of `value` with the forward's host read of num_rendered removed (gsr_forward_async inside gsrast.rasterize.static_capacity())"}
    if rank == 0 and not args.profile_all_stages_in_timed_region:
        gsrast.profile_enable(True)
        for _ in range(max(1, args.stage_steps)):
            step()
        torch.cuda.synchronize(device)
        prof = gsrast.profile_read()
        gsrast.profile_enable(False)
        prof["blend_bwd_timed_region"] = prof_timed["blend_bwd"]
    graph_info = None
    if rank == 0 and world == 1 and not args.no_graph_replay and os.environ.get("GSR_BENCH_TORCH_ADAM", "0") == "1":
        graph_info = {"skipped": "GSR_BENCH_TORCH_ADAM=1: torch's fused Adam is built with capturable=False and cannot be recorded"}
    elif rank == 0 and world == 1 and not args.no_graph_replay:
        # informational: the SAME step recorded once into a HIP graph (sync-free rasterizer forward, Adam scalars from device memory) and replayed;
        # `value` above stays the eagerly launched loop, whose dominant kernel is timed live with HIP events as the contract asks.  Run in a child
        # process: nothing that happens while recording / replaying a graph can take the headline line down with it.
        import subprocess
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--graph-replay-child", "--variant", args.variant, "--P", str(args.P), "--W", str(args.W),
                   "--H", str(args.H), "--color-mode", args.color_mode, "--steps", str(args.steps), "--warmup", str(args.warmup)]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=child_env())
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            graph_info = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": f"child rc={r.returncode}: {r.stderr[-200:]}"}
        except Exception as e:
            graph_info = {"error": str(e)[:200]}
    if rank == 0:
        from gsrast import runner as hiprun
        st = hiprun.run_raw(args.variant, sc, device=device)
        R = int(st["R"])
        R_after = int(hiprun.run_raw(args.variant, state["current_scene"](), device=device)["R"])
        N = args.W * args.H
        T = ((args.W + 15) // 16) * ((args.H + 15) // 16)
        fwd_b, bwd_b = algorithmic_bytes(args.variant, R, N, T)
        ms = {k: (v[0] / max(v[1], 1)) for k, v in prof.items()}
        if "blend_bwd_timed_region" in ms:          # the dominant kernel's figure is the one measured inside the timed region
            ms["blend_bwd_stage_pass"] = ms["blend_bwd"]; ms["blend_bwd"] = ms.pop("blend_bwd_timed_region")
        dom, dom_bytes = ("blend_bwd", bwd_b) if ms["blend_bwd"] >= ms["blend_fwd"] else ("blend_fwd", fwd_b)
        achieved = dom_bytes / (ms[dom] * 1e-3) / 1e9 if ms[dom] > 0 else 0.0
        bwd_sp = not os.environ.get("GSR_BWD", "sp").startswith("p")      # which backward formulation the library runs (gsr_blend.hip)
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get(f"{args.variant}:{dom}" + ("_sp" if (dom == "blend_bwd" and bwd_sp) else ""))
            except Exception:
                traffic = None
        raster_fwd = ms["preprocess"] + ms["depth_order"] + ms["binning"] + ms["blend_fwd"]
        raster_bwd = ms["bwd_memset"] + ms["blend_bwd"] + ms["preprocess_bwd"]
        # informational second ceiling for the dominant kernel: VALU issue rate.  Instruction count per launch from the committed PMC pass of
        # the same workload (profiles/r01_pmc_summary.json, SQ_INSTS_VALU; the counter was calibrated against a kernel of known instruction
        # count, tools/microbench/valu_count.py: +0.15 %); duration measured live.  Peak: CDNA4 CUs have four SIMD-32 units, so a wave64
        # VALU instruction issues in 2 cycles: 256 CU x 4 SIMD x 2.4 GHz / 2 = 1228.8 G wave-instructions/s (= the 157.3 TFLOP/s fp32 vector
        # peak of MI355X_MICROARCH.md / 128 flop per wave64 FMA).  tools/microbench/valu_ops.py measures 1020 G/s for independent v_mov_b32
        # and 810 G/s for dependent v_fma_f32 chains on this device.  tools/microbench/valu_rate.hip (profiles/r02_valu_issue_microbench.txt)
        # measures the ceilings by operand kind: only VGPR-only fma / mul / add reach ~890 G/s; any DPP form, any SGPR operand and
        # v_min / v_max issue at ~575 G/s, transcendentals at ~300 G/s -- the instruction mix of these kernels caps them well below 1228.8.
        valu = None
        pj = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json")) if os.path.exists(q)), "")
        vidx = {"ewa": 0, "surfel": 1, "plane": 2}[args.variant]
        if os.path.exists(pj) and args.P == 300000 and (args.W, args.H) == (1920, 1080) and args.color_mode == "precomp":
            try:
                kname = (f"k_blend_bwd_sp<{vidx}>" if bwd_sp else f"k_blend_bwd<{vidx}>") if dom == "blend_bwd" else f"k_blend_fwd<{vidx}>"
                pmc = json.load(open(pj))
                insts = pmc.get(kname, {}).get("SQ_INSTS_VALU")
                if insts:
                    rate = insts / (ms[dom] * 1e-3)
                    valu = {"wave_insts_per_launch": int(insts), "achieved_Ginst_s": round(rate / 1e9, 1), "peak_Ginst_s": 1228.8,
                            "frac": round(rate / 1228.8e9, 4),
                            "measured_ceilings_Ginst_s": {"vgpr_only_fma": 890, "dpp_or_sgpr_operand": 575, "transcendental": 300}, "source": f"profiles/{os.path.basename(pj)} SQ_INSTS_VALU (PMC pass of git revision "
                                      f"{pmc.get('_meta', {}).get('git_revision_of_the_measured_library', 'unrecorded')}) / live avg_launch_ms"}
            except Exception:
                valu = None
        out = {
            "metric": "train iters/sec @300k Gaussians 1080p (rasterize fwd+bwd ms and HBM GB/s vs roofline alongside)",
            "value": round(total_iters / elapsed, 3), "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "sync_free_forward": bool(args.sync_free), "sync_free": sync_free_info, "backend": backend, "dist_world_size": (dist.get_world_size() if dist is not None else 1), "ranks": ranks_seen,
            "distinct_devices": (len({(r["uuid"], r["pci"]) for r in ranks_seen}) if ranks_seen else 1),
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"diff_{ {'ewa':'gaussian','surfel':'surfel','plane':'plane'}[args.variant] }_rasterization forward+backward "
                                   f"({ {'ewa':'3DGS EWA','surfel':'2DGS surfel: the rasterizer of configs[1] scaffold-2dgs','plane':'PGSR plane: the rasterizer of configs[2]'}[args.variant] }) "
                                   f"+ fused L1/linear image loss + fused Adam (gsrast.optim, one HIP kernel) on EXPLICIT Gaussians, {'SH deg 3' if args.color_mode == 'sh' else 'colors_precomp'}, "
                                   f"P={args.P}, {args.W}x{args.H}, synthetic scene SURVEY §8d (seed=rank); the complete methods (decode, SSIM, "
                                   f"regularisers, statistics) are timed in method_iteration",
                       "variant": args.variant, "P": args.P, "W": args.W, "H": args.H, "tile_instances_R": R, "tile_instances_R_after_timed_steps": R_after,
                       "visible": int((st["radii"] > 0).sum()),
                       "clock_prewarm_ms": round(prewarm_ms, 1),
                       **({"skew": {"frac": args.skew_frac, "scale": args.skew_scale, "what": "NOT the BASELINE workload: gaussians concentrated at the image centre"}} if args.skew_frac > 0 else {}),
                       "depth_order": "global 4-pass radix sort of the gaussians" if depth_order_is_global(args.P, T, args.variant) else ("per-tile sort of the binned lists, in k_blend_fwd's prologue" if os.environ.get("GSR_TILE_SORT", "fused")[:1] != "k" else "per-tile sort of the binned lists (k_tile_depth_sort)"),
                       "tiles": T, "tiles_touched": int((st["ranges"][:, 1] > st["ranges"][:, 0]).sum()),
                       "gaussians_per_tile_mean": round(R / max(int((st["ranges"][:, 1] > st["ranges"][:, 0]).sum()), 1), 1),
                       "gaussians_per_tile_max": int((st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0].astype(np.int64)).max()),
                       "parallelism": f"{world} independent tile(s), 1 per GPU, no collective"},
            "rasterize_fwd_ms": round(raster_fwd, 4), "rasterize_bwd_ms": round(raster_bwd, 4),
            "rasterize_fwd_bwd_ms": round(raster_fwd + raster_bwd, 4),
            "stage_ms": {k: round(v, 4) for k, v in ms.items()},
            "stage_ms_note": ("all stages timed with HIP events inside the timed region" if args.profile_all_stages_in_timed_region else
                              f"blend_bwd: HIP events inside the timed region ({args.steps} launches); the other stages (and blend_bwd_stage_pass): a separate "
                              f"pass of {max(1, args.stage_steps)} iterations after it, every stage timed"),
            "stage_algorithmic_GBps": {k: round(b / (ms[k] * 1e-3) / 1e9, 1) for k, b in
                                       stage_bytes(args.variant, args.color_mode, args.P, R, N, T).items() if ms.get(k, 0) > 0},
            "roofline": {"kernel": (f"k_blend_bwd_sp<{args.variant}>" if bwd_sp else f"k_blend_bwd<{args.variant}>") if dom == "blend_bwd" else f"k_blend_fwd<{args.variant}>", "bound": "hbm",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(ms[dom], 4),
                         "valu_issue": valu,
                         "note": "blend is VALU/atomic-bound by construction (SURVEY §7-5); HBM fraction reported as BASELINE asks"},
        }
        try:
            pm = measured_copy_gbs(device)
            out["roofline"].update(peak_measured=round(pm, 1), frac_of_measured=round(achieved / pm, 5),
                                   peak_measured_what="1 GiB float32 device-to-device copy, read+write bytes / time")
        except Exception:
            pass
        if graph_info is not None:
            out["graph_replay"] = graph_info
        if world == 1 and not args.no_method_iteration and args.variant == "surfel" and (args.W, args.H) == (1920, 1080):
            # every method in its own child process (its pipelines, profiler session and graph capture cannot disturb the headline line)
            import subprocess
            out["method_iteration"] = {}
            for m in ("scaffold-2dgs", "octree-2dgs", "octree-pgsr", "pgsr"):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--method-iteration-child", m], capture_output=True, text=True, timeout=600,
                                       env=child_env())
                    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    out["method_iteration"][m] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": f"child rc={r.returncode}: {r.stderr[-200:]}"}
                except Exception as e:
                    out["method_iteration"][m] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            og = scenes.random_out_grads(args.variant, args.W, args.H, seed=0)
            out["cpu_baseline"], kept = cpu_baseline(args.variant, sc, og)
            ev, co = kept["pairs"]
            out["config"].update(pairs_evaluated_by_the_reference_loop=ev, contributing_pairs_C=co,
                                 contributing_pairs_per_tile_instance=round(co / max(R, 1), 2))
            out["parity_full_size"] = parity_full_size(args.variant, sc, og, device, args.color_mode)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
