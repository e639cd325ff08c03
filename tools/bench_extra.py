"""HBM-bound adjacent kernels on the GPU box: TSDF integrate (both definitions), distCUDA2, visible_filter.  Prints JSON."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes, hiprun
from gsrast.tsdf import tsdf_integrate_, DenseTSDFVolume
from simple_knn._C import distCUDA2
import scaffold_filter


def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


out = {}
W, H = 1920, 1080
cam = scenes.make_camera(W, H, 1600.0, 1600.0)
depth = torch.rand(1, H, W, device="cuda") * 2 + 4; rgb = torch.rand(3, H, W, device="cuda")
n = 512
g = torch.stack(torch.meshgrid(torch.linspace(-3, 3, n), torch.linspace(-1.7, 1.7, n), torch.linspace(3.5, 6.5, n), indexing="ij"), -1).reshape(-1, 3).cuda().contiguous()
V = g.shape[0]
ts = torch.ones(V, device="cuda"); ws = torch.ones(V, device="cuda"); cs = torch.zeros(V, 3, device="cuda")
F = torch.from_numpy(cam["projmatrix"]).cuda()
t = timeit(lambda: tsdf_integrate_(g, F, depth, rgb, 0.05, ts, cs, ws))
touched = float((ws > 1).float().mean())
alg = V * 12 + touched * V * 40
out["tsdf_points"] = {"voxels": V, "ms": round(t * 1e3, 3), "touched_frac": round(touched / 1.0, 3), "algorithmic_GBps": round(alg / t / 1e9, 1)}
vol = DenseTSDFVolume((-3, -1.7, 3.5), 6.0 / n, (n, n // 2, n), 0.05)
E = np.eye(4, dtype=np.float32)
t = timeit(lambda: vol.integrate(rgb, depth, 1600.0, 1600.0, W / 2, H / 2, E, quantize_rgb8=False))
Vd = n * (n // 2) * n
tf = float((vol.weight > 0).float().mean())
out["tsdf_dense"] = {"voxels": Vd, "ms": round(t * 1e3, 3), "touched_frac": round(tf, 3), "algorithmic_GBps": round((tf * Vd * 40) / t / 1e9, 1),
                     "streamed_GBps_lower_bound": round((Vd * 4) / t / 1e9, 1)}
sc = scenes.make_scene("ewa", 300000, W, H, seed=0)
pts = torch.from_numpy(sc["means3D"]).cuda()
t = timeit(lambda: distCUDA2(pts), n=5, w=1)
out["distCUDA2_300k"] = {"ms": round(t * 1e3, 3)}
td = hiprun.to_dev(sc)
rs = scaffold_filter.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=td["bg"],
        scale_modifier=1.0, viewmatrix=td["viewmatrix"], projmatrix=td["projmatrix"], sh_degree=0, campos=td["campos"], prefiltered=False, debug=False)
rr = scaffold_filter.GaussianRasterizer(rs)
t = timeit(lambda: rr.visible_filter(td["means3D"], td["scales"], td["rotations"]), n=50)
out["visible_filter_300k"] = {"ms": round(t * 1e3, 4), "algorithmic_GBps": round(300000 * 44 / t / 1e9, 1)}
# fused L1+SSIM photometric loss (value + dL/dimg) vs the reference's torch formula, 3x1080x1920
from gsrast.losses import l1_ssim
import ref_loss_torch
img = torch.rand(3, H, W, device="cuda").requires_grad_(True); gt = torch.rand(3, H, W, device="cuda")
def hip_loss():
    img.grad = None; l1_ssim(img, gt, 0.2).backward()
def torch_loss():
    img.grad = None; ref_loss_torch.loss(img.unsqueeze(0), gt.unsqueeze(0), 0.2)[0].backward()
th, tt = timeit(hip_loss, n=30), timeit(torch_loss, n=10)
npx = 3 * H * W
out["l1_ssim_1080p"] = {"hip_fwd_bwd_ms": round(th * 1e3, 3), "torch_fwd_bwd_ms": round(tt * 1e3, 3), "speedup": round(tt / th, 1),
                        "algorithmic_GBps": round(npx * 44 / th / 1e9, 1)}
# fused 2DGS geometric regularisers (render() post-processing + depth_to_normal + normal/dist losses) vs the torch chain
from gsrast.losses import camera_ray_matrices, surfel_geo_loss
import ref_geo_torch, test_loss_cpu
am, camg = test_loss_cpu._geo_case(H, W, 9)
wvt = torch.tensor(camg["viewmatrix"]).cuda(); fpt = torch.tensor(camg["projmatrix"]).cuda()
rm, nr = camera_ray_matrices(wvt, fpt, W, H)
xa = torch.tensor(am).cuda().requires_grad_(True)
def hip_geo():
    xa.grad = None; surfel_geo_loss(xa, rm, nr, 0.0, 0.05, 100.0)[0].backward()
def torch_geo():
    xa.grad = None; ref_geo_torch.geo_loss(xa, wvt, fpt, 0.0, 0.05, 100.0)[0].backward()
th, tt = timeit(hip_geo, n=30), timeit(torch_geo, n=10)
out["surfel_geo_loss_1080p"] = {"hip_fwd_bwd_ms": round(th * 1e3, 3), "torch_fwd_bwd_ms": round(tt * 1e3, 3), "speedup": round(tt / th, 1),
                                "algorithmic_GBps": round(H * W * 72 / th / 1e9, 1)}
# PGSR multi-view losses (geometric consistency + 102400 x 7x7 patch NCC), value + gradients, 1080p, vs the torch op chain
import mv_cases, ref_mv_torch
from gsrast.losses import multiview_cfg, plane_multiview_loss
mc = mv_cases.plane_pair(W=W, H=H, seed=11, amp=0.002, tex=25.0)
tt_ = lambda a: torch.tensor(a, device="cuda")
mnames = ("plane_depth", "near_plane_depth", "rendered_normal", "rendered_distance")
ml = {k: tt_(mc[k]).requires_grad_(True) for k in mnames}
mg, mng = tt_(mc["gray"]), tt_(mc["near_gray"])
mcfg = multiview_cfg(mv_cases.cam_ns(mc["view"]), mv_cases.cam_ns(mc["near"]), W, H, near_size=(W, H))
def hip_mv():
    for v in ml.values(): v.grad = None
    a, b = plane_multiview_loss(*[ml[k] for k in mnames], mg, mng, mcfg)
    (a + b).backward()
fixed_idx = plane_multiview_loss(*[ml[k] for k in mnames], mg, mng, mcfg, return_aux=True)[2]["indices"]
def hip_mv_fixed():
    for v in ml.values(): v.grad = None
    a, b = plane_multiview_loss(*[ml[k] for k in mnames], mg, mng, mcfg, indices=fixed_idx)
    (a + b).backward()
def torch_mv():
    for v in ml.values(): v.grad = None
    a, b = ref_mv_torch.multiview_loss(*[ml[k] for k in mnames], mg, mng, mc["view"], mc["near"], indices=fixed_idx)
    (a + b).backward()
th, thf, tt = timeit(hip_mv, n=20), timeit(hip_mv_fixed, n=20), timeit(torch_mv, n=5, w=2)
out["plane_multiview_1080p"] = {"hip_fwd_bwd_ms": round(th * 1e3, 3), "hip_fwd_bwd_ms_given_indices": round(thf * 1e3, 3), "torch_fwd_bwd_ms": round(tt * 1e3, 3),
                                "speedup": round(tt / th, 1), "samples": 102400, "note": "torch chain timed WITHOUT its host-side np.random.choice / nonzero"}
# per-Gaussian all_map input of the plane rasterizer (PGSRScene.render prep) vs its torch op chain, P = 300k
from gsrast.plane_prep import plane_input_all_map
Pp = 300000
px_ = (torch.rand(Pp, 3, device="cuda") * 10 - 5).requires_grad_(True); pq_ = torch.nn.functional.normalize(torch.randn(Pp, 4, device="cuda")).requires_grad_(True)
ps_ = torch.exp(torch.randn(Pp, 3, device="cuda") - 2); pg_ = torch.randn(Pp, 5, device="cuda")
Vm = torch.tensor(camg["viewmatrix"]).cuda(); cpos = torch.linalg.inv(Vm)[3, :3].contiguous()
def q2m(q):
    r, i, j, k = torch.unbind(q, -1); two_s = 2.0 / (q * q).sum(-1)
    return torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r), two_s * (i * j + k * r), 1 - two_s * (i * i + k * k),
                        two_s * (j * k - i * r), two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1).reshape(-1, 3, 3)
def torch_prep():
    px_.grad = None; pq_.grad = None
    R = q2m(pq_); idx = ps_.min(dim=-1)[1][..., None, None].expand(-1, 3, -1)
    n = R.gather(2, idx).squeeze(2)
    neg = (n * (cpos - px_)).sum(-1) < 0.0
    n = torch.where(neg[:, None], -n, n)
    ln = n @ Vm[:3, :3]; pc = px_ @ Vm[:3, :3] + Vm[3, :3]
    am = torch.zeros(Pp, 5, device="cuda"); am[:, :3] = ln; am[:, 3] = 1.0; am[:, 4] = (ln * pc).sum(-1).abs()
    (am * pg_).sum().backward()
def hip_prep():
    px_.grad = None; pq_.grad = None
    (plane_input_all_map(px_, pq_, ps_, Vm, cpos) * pg_).sum().backward()
th, tt = timeit(hip_prep, n=30), timeit(torch_prep, n=10)
out["plane_allmap_prep_300k"] = {"hip_fwd_bwd_ms": round(th * 1e3, 3), "torch_fwd_bwd_ms": round(tt * 1e3, 3), "speedup": round(tt / th, 1),
                                 "note": "both include the (am * g).sum().backward() driver ops"}
# per-iteration densification statistics of the Scaffold / Octree methods (training_statis) vs its torch chain: 72k anchors x 10 offsets
import ref_decode_torch
from gsrast.decode import training_stats_, compact_visible
Na_, k_ = 72000, 10
gen_ = torch.Generator(device="cuda").manual_seed(5)
vis_ = torch.rand(Na_, device="cuda", generator=gen_) < 0.88; Nv_ = int(vis_.sum())
nop_ = torch.tanh(torch.randn(Nv_ * k_, device="cuda", generator=gen_)); sel_ = nop_ > 0; P_ = int(sel_.sum())
upd_ = torch.rand(P_, device="cuda", generator=gen_) < 0.8; vg_ = torch.randn(P_, 3, device="cuda", generator=gen_)
acc_ = {"opacity_accum": torch.zeros(Na_, 1, device="cuda"), "anchor_demon": torch.zeros(Na_, 1, device="cuda"),
        "offset_gradient_accum": torch.zeros(Na_ * k_, 1, device="cuda"), "offset_denom": torch.zeros(Na_ * k_, 1, device="cuda")}
vi_ = compact_visible(vis_)
th = timeit(lambda: training_stats_(acc_["opacity_accum"], acc_["anchor_demon"], acc_["offset_gradient_accum"], acc_["offset_denom"], vg_, nop_, upd_, sel_,
                                    vis_idx=vi_), n=50)
tt = timeit(lambda: ref_decode_torch.training_statis(acc_, k_, vg_, nop_.view(-1, 1), upd_, sel_, vis_), n=20)
out["scaffold_training_stats_72k_anchors"] = {"hip_ms": round(th * 1e3, 4), "torch_ms": round(tt * 1e3, 3), "speedup": round(tt / th, 1), "P": P_,
                                              "note": "the torch chain's boolean-mask indexing synchronises with the host six times per call"}
print(json.dumps(out))
