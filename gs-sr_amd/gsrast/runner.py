"""Runs a gsrast.workloads scene through the PRODUCT (drop-in python packages -> C ABI -> HIP kernels)."""
import numpy as np
import torch

import diff_gaussian_rasterization as dgr
import diff_surfel_rasterization as dsr
import diff_plane_rasterization as dpr
from gsrast import rasterize as rz
import gsrast

VID = {"ewa": gsrast.EWA, "surfel": gsrast.SURFEL, "plane": gsrast.PLANE}


def to_dev(sc, device="cuda"):
    t = {}
    for k, v in sc.items():
        if isinstance(v, np.ndarray):
            t[k] = torch.from_numpy(v).to(device)
        else:
            t[k] = v
    return t


def settings(variant, t, debug=False):
    kw = dict(image_height=int(t["H"]), image_width=int(t["W"]), tanfovx=float(t["tanfovx"]), tanfovy=float(t["tanfovy"]),
              bg=t["bg"], scale_modifier=float(t.get("scale_modifier", 1.0)), viewmatrix=t["viewmatrix"],
              projmatrix=t["projmatrix"], sh_degree=int(t.get("sh_degree", 0)), campos=t["campos"], prefiltered=False,
              debug=debug)
    if variant == "plane":
        return dpr.GaussianRasterizationSettings(render_geo=bool(t.get("render_geo", True)), **kw)
    if variant == "surfel":
        return dsr.GaussianRasterizationSettings(**kw)
    return dgr.GaussianRasterizationSettings(**kw)


def run(variant, sc, og=None, device="cuda", debug=False):
    """Forward (+ backward when og is given) through the public drop-in API.  Returns dict of numpy arrays."""
    t = to_dev(sc, device)
    P = t["means3D"].shape[0]
    leaves = {}
    for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp", "all_map"):
        if t.get(k) is not None:
            leaves[k] = t[k].clone().requires_grad_(og is not None)
    means2D = torch.zeros((P, 3), dtype=torch.float32, device=device, requires_grad=og is not None)
    rs = settings(variant, t, debug)
    kw = dict(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves.get("shs"),
              colors_precomp=leaves.get("colors_precomp"), scales=leaves.get("scales"), rotations=leaves.get("rotations"),
              cov3D_precomp=leaves.get("cov3D_precomp"))
    out = {}
    if variant == "ewa":
        color, radii = dgr.GaussianRasterizer(rs)(**kw)
    elif variant == "surfel":
        color, radii, others = dsr.GaussianRasterizer(rs)(**kw)
        out["others"] = others
    else:
        means2D_abs = torch.zeros((P, 3), dtype=torch.float32, device=device, requires_grad=og is not None)
        color, radii, observe, out_all_map, plane_depth = dpr.GaussianRasterizer(rs)(
            means2D_abs=means2D_abs, all_map=leaves.get("all_map"), **kw)
        out.update(observe=observe, out_all_map=out_all_map, plane_depth=plane_depth)
    out.update(color=color, radii=radii)
    res = {k: v.detach().cpu().numpy() for k, v in out.items()}
    if og is not None:
        loss = (color * torch.from_numpy(og["dL_dcolor"]).to(device)).sum()
        if variant == "surfel" and og.get("dL_dothers") is not None:
            loss = loss + (out["others"] * torch.from_numpy(og["dL_dothers"]).to(device)).sum()
        if variant == "plane":
            if og.get("dL_dout_all_map") is not None:
                loss = loss + (out["out_all_map"] * torch.from_numpy(og["dL_dout_all_map"]).to(device)).sum()
            if og.get("dL_dplane_depth") is not None:
                loss = loss + (out["plane_depth"] * torch.from_numpy(og["dL_dplane_depth"]).to(device)).sum()
        loss.backward()
        g = {"dL_d" + k: (v.grad.detach().cpu().numpy() if v.grad is not None else None) for k, v in leaves.items()}
        g["dL_dmeans2D"] = means2D.grad.detach().cpu().numpy()
        if variant == "plane":
            g["dL_dmeans2D_abs"] = means2D_abs.grad.detach().cpu().numpy()
        res["grads"] = g
    torch.cuda.synchronize()
    return res


def run_raw(variant, sc, device="cuda"):
    """Forward through gsrast.rasterize.forward, returning the integer stage results too (debug reads)."""
    t = to_dev(sc, device)
    rs = settings(variant, t)
    vid = VID[variant]
    R, outs, radii, geom, binning, img = rz.forward(vid, t["means3D"], t.get("shs"), t.get("colors_precomp"), t["opacities"],
                                                    t.get("scales"), t.get("rotations"), t.get("cov3D_precomp"),
                                                    t.get("all_map") if variant == "plane" else None, rs)
    P = t["means3D"].shape[0]
    W, H = int(t["W"]), int(t["H"])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    M = t["shs"].shape[1] if t.get("shs") is not None else 0
    dev = t["means3D"].device
    k = 3 if variant == "surfel" else 1
    k2 = 2 if variant == "surfel" else 1
    def rd(field, shape, dtype):
        o = torch.zeros(shape, dtype=dtype, device=dev)
        return rz.debug_read(vid, field, rs, P, M, R, geom, binning, img, o).cpu().numpy()
    st = dict(R=R, radii=radii.cpu().numpy(),
              tiles_touched=rd(0, (P,), torch.int32).view(np.uint32),
              point_list=rd(1, (max(R, 1),), torch.int32).view(np.uint32)[:R],
              tile_keys=rd(5, (max(R, 1),), torch.int32).view(np.uint32)[:R],
              ranges=rd(2, (gx * gy, 2), torch.int32).view(np.uint32),
              final_T=rd(3, (k, H, W), torch.float32),
              n_contrib=rd(4, (k2, H, W), torch.int32).view(np.uint32))
    st.update({kk: v.cpu().numpy() for kk, v in outs.items()})
    torch.cuda.synchronize()
    return st
