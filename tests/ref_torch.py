"""float64 PyTorch-autograd restatement of the FORWARD maths of the three rasterizers (dense pixels x gaussians).

TEST INFRASTRUCTURE ONLY.  Purpose: pin the oracle's hand-derived analytic backward (oracle/gsr_oracle.c, which
follows the reference's backward.cu) against automatic differentiation of an independently written forward.
Small sizes only (memory is O(pixels * gaussians)).

Reference semantics that are NOT plain calculus are reproduced on purpose so autograd matches the reference:
  * no gradient through the min(0.99, .) clamp's *test* (3DGS backward.cu:498-499,538) -> straight-through,
  * discrete gates (alpha<1/255, power>0, T<1e-4, tile rect, near plane) are constants,
  * means2D is a dummy zero input whose gradient is dL/d(pixel mean) * 0.5*(W,H) (3DGS backward.cu:460-461,545-546).
Quirks that autograd cannot express (surfel median-normal to all splats, surfel scale_modifier ignored in backward,
surfel means2D densification proxy, tan-fov clamp, PLANE abs-gradient) are tested separately in tests/.
"""
import math
import numpy as np
import torch

F64 = torch.float64
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def eval_sh_color(deg, sh, dirs):
    """3DGS forward.cu:20-71 (without the clamp)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                 + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                r = (r + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                     + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                     + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                     + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                     + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r + 0.5


def quat_R(q, normalize):
    if normalize:
        q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    return R   # standard rotation matrix, columns = local axes


def _rect(cx, cy, radius, gx, gy):
    """getRect (3DGS auxiliary.h:46-56) with C int truncation."""
    def tr(v):
        return torch.trunc(v).to(torch.int64)
    r = radius.to(F64)
    x0 = torch.clamp(tr((cx - r) / 16), 0, gx)
    y0 = torch.clamp(tr((cy - r) / 16), 0, gy)
    x1 = torch.clamp(tr((cx + r + 15) / 16), 0, gx)
    y1 = torch.clamp(tr((cy + r + 15) / 16), 0, gy)
    return x0, y0, x1, y1


def render(variant, sc, leaves=None, pair_xy_leaf=False, surfel_quat_jacobian=False):
    """sc: scene dict (numpy, tests/scenes.py).  Returns (outputs dict, leaves dict of float64 tensors w/ grad)."""
    t = lambda a: torch.tensor(np.asarray(a), dtype=F64)
    W, H = int(sc["W"]), int(sc["H"])
    P = sc["means3D"].shape[0]
    if leaves is None:
        leaves = {}
        for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp", "shs", "all_map"):
            if sc.get(k) is not None and (k != "all_map" or variant == "plane"):
                leaves[k] = t(sc[k]).requires_grad_(True)
        leaves["means2D"] = torch.zeros(P, 3, dtype=F64, requires_grad=True)
    L = leaves
    V, Fm = t(sc["viewmatrix"]), t(sc["projmatrix"])
    campos, bg = t(sc["campos"]), t(sc["bg"])
    tanx, tany = float(np.float32(sc["tanfovx"])), float(np.float32(sc["tanfovy"]))
    mod = float(sc.get("scale_modifier", 1.0))
    fx = float(np.float32(W) / (np.float32(2.0) * np.float32(tanx)))
    fy = float(np.float32(H) / (np.float32(2.0) * np.float32(tany)))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    means = L["means3D"]
    opac = L["opacities"].reshape(-1)

    p_view = means @ V[:3, :3] + V[3, :3]
    vis = p_view[:, 2] > 0.2
    ones = torch.ones(P, 1, dtype=F64)
    p_hom = torch.cat([means, ones], 1) @ Fm

    if variant in ("ewa", "plane"):
        p_w = 1.0 / (p_hom[:, 3] + 1e-7)
        ndc = p_hom[:, :2] * p_w[:, None] + L["means2D"][:, :2]
        cx = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
        cy = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
        R = quat_R(L["rotations"], normalize=False)
        S = mod * L["scales"]
        Mm = R * S[:, None, :]
        Sigma = Mm @ Mm.transpose(1, 2)
        tx, ty, tz = p_view[:, 0], p_view[:, 1], p_view[:, 2]
        limx, limy = 1.3 * tanx, 1.3 * tany
        txc = torch.clamp(tx / tz, -limx, limx) * tz
        tyc = torch.clamp(ty / tz, -limy, limy) * tz
        J = torch.zeros(P, 2, 3, dtype=F64)
        J[:, 0, 0] = fx / tz; J[:, 0, 2] = -(fx * txc) / (tz * tz)
        J[:, 1, 1] = fy / tz; J[:, 1, 2] = -(fy * tyc) / (tz * tz)
        Rw2c = V[:3, :3].T            # p_view = R_w2c p + t
        JW = J @ Rw2c
        cov = JW @ Sigma @ JW.transpose(1, 2)
        a = cov[:, 0, 0] + 0.3; b = cov[:, 0, 1]; c = cov[:, 1, 1] + 0.3
        det = a * c - b * b
        vis = vis & (det != 0)
        conA, conB, conC = c / det, -b / det, a / det
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    else:
        # The reference's quat_to_rotmat_vjp (SURFEL auxiliary.h:241-284) differentiates the rotation formula at the
        # normalised quaternion WITHOUT the normalisation Jacobian; with unit-norm inputs that equals differentiating
        # the un-normalised formula, which is what surfel_quat_jacobian=False does.
        R = quat_R(L["rotations"], normalize=surfel_quat_jacobian)
        s2 = mod * L["scales"]
        L0 = R[:, :, 0] * s2[:, 0:1]; L1 = R[:, :, 1] * s2[:, 1:2]; L2 = R[:, :, 2]
        zeros = torch.zeros(P, 1, dtype=F64)
        rows = torch.stack([torch.cat([L0, zeros], 1), torch.cat([L1, zeros], 1), torch.cat([means, ones], 1)], 1)  # P,3,4
        Nm = torch.zeros(4, 3, dtype=F64)
        Nm[0, 0] = W / 2.0; Nm[3, 0] = (W - 1) / 2.0; Nm[1, 1] = H / 2.0; Nm[3, 1] = (H - 1) / 2.0; Nm[3, 2] = 1.0
        Tm = rows @ Fm @ Nm          # P,3(rows u,v,c),3(cols x,y,w)
        Tu, Tv, Tw = Tm[:, :, 0], Tm[:, :, 1], Tm[:, :, 2]
        normal = L2 @ V[:3, :3]
        cosv = -(p_view * normal).sum(-1)
        vis = vis & (cosv != 0)
        normal = normal * torch.where(cosv > 0, 1.0, -1.0).detach()[:, None]
        tvec = torch.tensor([9.0, 9.0, -1.0], dtype=F64)
        d = (tvec * Tw * Tw).sum(-1)
        vis = vis & (d != 0)
        f = tvec[None] / d[:, None]
        cx = (f * Tu * Tw).sum(-1); cy = (f * Tv * Tw).sum(-1)
        hx = torch.sqrt(torch.clamp(cx * cx - (f * Tu * Tu).sum(-1), min=1e-4))
        hy = torch.sqrt(torch.clamp(cy * cy - (f * Tv * Tv).sum(-1), min=1e-4))
        radius = torch.ceil(torch.maximum(torch.maximum(hx, hy), torch.tensor(3.0 * 0.707106, dtype=F64))).detach()

    x0, y0, x1, y1 = _rect(cx.detach(), cy.detach(), radius, gx, gy)
    vis = vis & (((x1 - x0) * (y1 - y0)) != 0)
    radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)

    if L.get("shs") is not None:
        dirs = means - campos
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        col = torch.clamp(eval_sh_color(int(sc.get("sh_degree", 0)), L["shs"], dirs), min=0.0)
    else:
        col = L["colors_precomp"]

    # global (depth, index) order == per-tile order of the reference's stable (tile|depth) sort
    dkey = p_view[:, 2].detach().to(torch.float32).numpy().view(np.uint32).astype(np.int64)
    order = torch.tensor(np.lexsort((np.arange(P), dkey)), dtype=torch.int64)
    order = order[vis[order]]
    G_ = order.numel()

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    px = xs.reshape(-1).to(F64); py = ys.reshape(-1).to(F64)
    ptx = (xs.reshape(-1) // 16); pty = (ys.reshape(-1) // 16)
    in_tile = ((ptx[:, None] >= x0[order][None]) & (ptx[:, None] < x1[order][None]) &
               (pty[:, None] >= y0[order][None]) & (pty[:, None] < y1[order][None]))
    Np = px.numel()

    cxo, cyo = cx[order], cy[order]
    if pair_xy_leaf:
        pair_xy = torch.zeros(Np, G_, 2, dtype=F64, requires_grad=True)
        L["_pair_xy"] = pair_xy
        dxm = cxo[None] + pair_xy[:, :, 0] - px[:, None]
        dym = cyo[None] + pair_xy[:, :, 1] - py[:, None]
    else:
        dxm = cxo[None] - px[:, None]; dym = cyo[None] - py[:, None]
    o = opac[order][None]
    depth_pix = None
    if variant in ("ewa", "plane"):
        power = -0.5 * (conA[order][None] * dxm * dxm + conC[order][None] * dym * dym) - conB[order][None] * dxm * dym
        valid = in_tile & (power <= 0)
        Gm = torch.exp(power)
    else:
        Tuo, Tvo, Two = Tu[order], Tv[order], Tw[order]
        k = px[:, None, None] * Two[None] - Tuo[None]
        l = py[:, None, None] * Two[None] - Tvo[None]
        pp = torch.cross(k, l, dim=-1)
        okz = pp[:, :, 2] != 0
        ppz = torch.where(okz, pp[:, :, 2], torch.ones_like(pp[:, :, 2]))
        sx, sy = pp[:, :, 0] / ppz, pp[:, :, 1] / ppz
        rho3d = sx * sx + sy * sy
        rho2d = 2.0 * (dxm * dxm + dym * dym)
        use3d = rho3d <= rho2d
        rho = torch.where(use3d, rho3d, rho2d)
        depth_pix = torch.where(use3d, sx * Two[None, :, 0] + sy * Two[None, :, 1] + Two[None, :, 2],
                                Two[None, :, 2].expand(Np, G_))
        power = -0.5 * rho
        valid = in_tile & okz & (depth_pix >= 0.2) & (power <= 0)
        Gm = torch.exp(power)
    a_raw = o * Gm
    alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()     # straight-through clamp
    valid = valid & (alpha.detach() >= 1.0 / 255.0)
    av = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - av
    Tincl = torch.cumprod(one_m, dim=1)
    Tbefore = torch.cat([torch.ones(Np, 1, dtype=F64), Tincl[:, :-1]], 1)
    stop = valid & ((Tbefore * one_m).detach() < 1e-4)
    stopped = torch.cumsum(stop.to(torch.int64), dim=1) > 0
    contrib = valid & ~stopped
    ac = torch.where(contrib, alpha, torch.zeros_like(alpha))
    Tincl = torch.cumprod(1.0 - ac, dim=1)
    Tbefore = torch.cat([torch.ones(Np, 1, dtype=F64), Tincl[:, :-1]], 1)
    w = ac * Tbefore
    Tfinal = Tincl[:, -1] if G_ > 0 else torch.ones(Np, dtype=F64)
    out = {}
    color = w @ col[order] + Tfinal[:, None] * bg[None]
    out["color"] = color.T.reshape(3, H, W)
    out["radii"] = radii
    # n_contrib (position in the per-tile list) is compared through the images only
    if variant == "surfel":
        near, far = 0.2, 100.0
        dsafe = torch.where(contrib, depth_pix, torch.ones_like(depth_pix))
        m = far / (far - near) * (1 - near / dsafe)
        mw, m2w = m * w, m * m * w
        M1ex = torch.cumsum(mw, 1) - mw
        M2ex = torch.cumsum(m2w, 1) - m2w
        A = 1 - Tbefore
        dist = ((m * m * A + M2ex - 2 * m * M1ex) * w).sum(1)
        Dm = (dsafe * w).sum(1)
        Nmap = w @ normal[order]
        med_mask = contrib & (Tbefore.detach() > 0.5)
        # last contributor with incoming T > 0.5
        idxs = torch.arange(G_)[None].expand(Np, G_)
        last = torch.where(med_mask, idxs, torch.full_like(idxs, -1)).max(1).values
        has = last >= 0
        lastc = torch.clamp(last, min=0)
        med_depth = torch.where(has, dsafe.gather(1, lastc[:, None])[:, 0], torch.zeros(Np, dtype=F64))
        med_normal = torch.where(has[:, None], normal[order][lastc], torch.zeros(Np, 3, dtype=F64))
        med_idx = torch.where(has, order[lastc].to(F64), torch.full((Np,), -1.0, dtype=F64))
        others = torch.stack([Dm, 1 - Tfinal, Nmap[:, 0], Nmap[:, 1], Nmap[:, 2], med_depth, dist, med_idx,
                              med_normal[:, 0], med_normal[:, 1], med_normal[:, 2]], 0)
        out["others"] = others.reshape(11, H, W)
    if variant == "plane":
        am = w @ L["all_map"][order]
        out["out_all_map"] = am.T.reshape(5, H, W)
        rx = (px - W * 0.5) / fx; ry = (py - H * 0.5) / fy
        out["plane_depth"] = (am[:, 4] / -(am[:, 0] * rx + am[:, 1] * ry + am[:, 2] + 1e-8)).reshape(1, H, W)
        obs = (contrib & (Tbefore.detach() > 0.5)).sum(0)
        observe = torch.zeros(P, dtype=torch.int64); observe[order] = obs
        out["observe"] = observe
    out["_order"] = order
    return out, L


def backward(variant, sc, og, pair_xy_leaf=False):
    """Returns (outputs, grads dict numpy) for upstream grads og (tests/scenes.random_out_grads)."""
    out, L = render(variant, sc, pair_xy_leaf=pair_xy_leaf)
    t = lambda a: torch.tensor(np.asarray(a), dtype=F64)
    loss = (out["color"] * t(og["dL_dcolor"])).sum()
    if variant == "surfel" and og.get("dL_dothers") is not None:
        g = t(og["dL_dothers"]).clone()
        loss = loss + (out["others"] * g).sum()
    if variant == "plane":
        if og.get("dL_dout_all_map") is not None:
            loss = loss + (out["out_all_map"] * t(og["dL_dout_all_map"])).sum()
        if og.get("dL_dplane_depth") is not None:
            loss = loss + (out["plane_depth"] * t(og["dL_dplane_depth"])).sum()
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in L.items()}
    return out, grads
