"""Torch restatement of PGSR's multi-view losses (gssr/scene/pgsr_scene.py:113-204 + helpers), device-agnostic, for timing the op chain
the fused kernels replace and for full-size parity.  Cameras are the dicts of mv_cases (R, T, Fx, Fy, Cx, Cy).  TEST INFRASTRUCTURE."""
import torch
import torch.nn.functional as F


def _rays(cam, W, H, dev):
    ix, iy = torch.meshgrid(torch.arange(W, device=dev), torch.arange(H, device=dev), indexing="xy")
    return torch.stack([(ix - cam["Cx"]) / cam["Fx"], (iy - cam["Cy"]) / cam["Fy"], torch.ones_like(ix)], -1).float()


def _lncc(ref, nea):
    tps = nea.shape[1]
    rs, ns = ref.sum(1), nea.sum(1)
    r2, n2, rn = (ref * ref).sum(1), (nea * nea).sum(1), (ref * nea).sum(1)
    ravg, navg = rs / tps, ns / tps
    cross = rn - navg * rs
    rvar = r2 - ravg * rs
    nvar = n2 - navg * ns
    ncc = torch.clamp(1 - cross * cross / (rvar * nvar + 1e-8), 0.0, 2.0)
    return ncc, ncc < 0.9


def multiview_loss(plane_depth, near_plane_depth, normal, distance, gray, near_gray, vc, nc, lambda_geo=0.03, lambda_ncc=0.15, patch=3, th=1.0,
                   indices=None):
    dev = plane_depth.device
    H, W = plane_depth.shape[-2:]
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    Rv, Tv, Rn, Tn = t(vc["R"]), t(vc["T"]), t(nc["R"]), t(nc["T"])
    ix, iy = torch.meshgrid(torch.arange(W, device=dev), torch.arange(H, device=dev), indexing="xy")
    pixels = torch.stack([ix, iy], -1).float()
    pts = (_rays(vc, W, H, dev) * plane_depth.squeeze()[..., None]).reshape(-1, 3)
    pts = (pts - Tv) @ Rv.t()
    q = pts @ Rn + Tn
    Hn, Wn = near_plane_depth.shape[-2:]
    proj = torch.stack([q[:, 0] * nc["Fx"] / q[:, 2] + nc["Cx"], q[:, 1] * nc["Fy"] / q[:, 2] + nc["Cy"]], -1)
    mask = (proj[:, 0] > 0) & (proj[:, 0] < Wn) & (proj[:, 1] > 0) & (proj[:, 1] < Hn) & (q[:, 2] > 0.1)
    grid = torch.stack([proj[:, 0] / ((Wn - 1) / 2) - 1, proj[:, 1] / ((Hn - 1) / 2) - 1], -1).view(1, -1, 1, 2)
    map_z = F.grid_sample(near_plane_depth.reshape(1, 1, Hn, Wn), grid, mode="bilinear", padding_mode="border", align_corners=True)[0, :, :, 0]
    q = q / q[:, 2:3] * map_z.squeeze()[..., None]
    back = ((q - Tn) @ Rn.t()) @ Rv + Tv
    bp = torch.stack([back[:, 0] * vc["Fx"] / back[:, 2] + vc["Cx"], back[:, 1] * vc["Fy"] / back[:, 2] + vc["Cy"]], -1)
    noise = torch.norm(bp - pixels.reshape(-1, 2), dim=-1)
    d_mask = mask & (noise < th)
    weights = (1.0 / torch.exp(noise)).detach()
    weights = torch.where(d_mask, weights, torch.zeros_like(weights))
    zero = plane_depth.sum() * 0
    if d_mask.sum() == 0:
        return zero, zero
    geo = lambda_geo * (weights * noise)[d_mask].mean()
    with torch.no_grad():
        valid = torch.nonzero(d_mask).squeeze(1) if indices is None else indices.long()
        w = weights[valid]
        off = torch.arange(-patch, patch + 1, device=dev)
        oy, ox = torch.meshgrid(off, off, indexing="ij")
        offsets = torch.stack([ox, oy], -1).view(1, -1, 2).float()
        ori = pixels.reshape(-1, 2)[valid].reshape(-1, 1, 2) + offsets
        Hg, Wg = gray.shape[-2:]
        g = ori.clone()
        g[..., 0] = 2 * g[..., 0] / (Wg - 1) - 1.0
        g[..., 1] = 2 * g[..., 1] / (Hg - 1) - 1.0
        ref_val = F.grid_sample(gray.reshape(1, 1, Hg, Wg), g.view(1, -1, 1, 2), align_corners=True).reshape(-1, offsets.shape[1])
        r = Rn.t() @ Rv
        tt = -r @ Tv + Tn
    n = normal.permute(1, 2, 0).reshape(-1, 3)[valid]
    d = distance.reshape(-1)[valid]
    Hm = r[None] - (tt[None, :, None] * n[:, None, :]) / d[:, None, None]
    Kn = t([[nc["Fx"], 0, nc["Cx"]], [0, nc["Fy"], nc["Cy"]], [0, 0, 1]])
    Kvi = t([[1 / vc["Fx"], 0, -vc["Cx"] / vc["Fx"]], [0, 1 / vc["Fy"], -vc["Cy"] / vc["Fy"]], [0, 0, 1]])
    Hk = Kn[None] @ Hm @ Kvi[None]
    homo = torch.cat([ori, torch.ones_like(ori[..., :1])], -1)
    gt_ = torch.einsum("bik,bpk->bpi", Hk, homo)
    grid2 = gt_[..., :2] / (gt_[..., 2:] + 1e-10)
    gx = 2 * grid2[..., 0] / (Wg - 1) - 1.0
    gy = 2 * grid2[..., 1] / (Hg - 1) - 1.0
    samp = F.grid_sample(near_gray.reshape(1, 1, Hg, Wg), torch.stack([gx, gy], -1).reshape(1, -1, 1, 2), align_corners=True).reshape(-1, offsets.shape[1])
    ncc, m = _lncc(ref_val, samp)
    ncc = ncc * w
    if m.sum() == 0:
        return geo, zero
    return geo, lambda_ncc * ncc[m].mean()
