"""float64 torch transcription of generate_neural_gaussians (gssr/scene/scaffold_scene.py:27-120, octree_scene.py:26-133) used to pin
the C oracle (forward values and autograd gradients).  TEST INFRASTRUCTURE ONLY.  The op sequence mirrors the reference's: boolean-mask
gathers, cat of [feat, ob_view, (ob_dist), (level)], three Sequential(Linear, ReLU, Linear, act) heads, concatenated masking, split."""
import torch
import torch.nn.functional as Fn


def _chain(case, leaves, par, vis, campos, lvl_t, osc_t, mask_override=None):
    k = case["k"]
    anchor = leaves["anchor"][vis]; feat = leaves["feat"][vis]
    grid_offsets = leaves["offset"][vis]; grid_scaling = leaves["scaling"][vis]
    ob_view = anchor - campos
    ob_dist = ob_view.norm(dim=1, keepdim=True)
    ob_view = ob_view / ob_dist
    lvl = [] if lvl_t is None else [lvl_t[vis].unsqueeze(1)]
    with_dist = torch.cat([feat, ob_view, ob_dist] + lvl, dim=1)
    wo_dist = torch.cat([feat, ob_view] + lvl, dim=1)
    pick = lambda flag: with_dist if flag else wo_dist

    def head(x, W1, b1, W2, b2):
        return Fn.linear(torch.relu(Fn.linear(x, par[W1], par[b1])), par[W2], par[b2])
    neural_opacity = torch.tanh(head(pick(case["dist_o"]), "W1o", "b1o", "W2o", "b2o"))
    if osc_t is not None:
        neural_opacity = neural_opacity * osc_t[vis].unsqueeze(1)
    neural_opacity = neural_opacity.reshape([-1, 1])
    mask = (neural_opacity > 0.0).view(-1) if mask_override is None else mask_override
    opacity = neural_opacity[mask]
    xk = pick(case["dist_k"])
    if par.get("app") is not None:
        xk = torch.cat([xk, par["app"].unsqueeze(0).expand(xk.shape[0], -1)], dim=1)
    color = torch.sigmoid(head(xk, "W1k", "b1k", "W2k", "b2k")).reshape([anchor.shape[0] * k, 3])
    scale_rot = head(pick(case["dist_c"]), "W1c", "b1c", "W2c", "b2c").reshape([anchor.shape[0] * k, 7])
    offsets = grid_offsets.view([-1, 3])
    concatenated = torch.cat([grid_scaling, anchor], dim=-1)
    rep = concatenated.unsqueeze(1).expand(-1, k, -1).reshape(-1, 9)          # repeat 'n c -> (n k) c'
    masked = torch.cat([rep, color, scale_rot, offsets], dim=-1)[mask]
    scaling_repeat, repeat_anchor, color, scale_rot, offsets = masked.split([6, 3, 3, 7, 3], dim=-1)
    scaling = scaling_repeat[:, 3:] * torch.sigmoid(scale_rot[:, :3])
    rot = Fn.normalize(scale_rot[:, 3:7])
    xyz = repeat_anchor + offsets * scaling_repeat[:, :3]
    return {"xyz": xyz, "color": color, "opacity": opacity.view(-1), "scaling": scaling, "rot": rot,
            "neural_opacity": neural_opacity.view(-1), "mask": mask}


def decode_live(case, leaves, par, vis, campos):
    """the chain on caller-owned device tensors (benchmark use)"""
    return _chain(case, leaves, par, vis, campos, None, None), leaves


def decode(case, dtype=torch.float64, mask_override=None, device="cpu"):
    """-> (outputs dict, leaves dict).  Leaves require grad; outputs are torch tensors (compacted like the reference)."""
    t = lambda a: None if a is None else torch.tensor(a, dtype=dtype, device=device)
    leaves = {n: t(case[n]).requires_grad_(True) for n in ("anchor", "feat", "offset", "scaling")}
    par = {n: t(v).requires_grad_(True) for n, v in case["params"].items() if v is not None}
    vis = torch.tensor(case["vis_idx"], dtype=torch.long, device=device)
    mo = None if mask_override is None else torch.tensor(mask_override, dtype=torch.bool, device=device)
    out = _chain(case, leaves, par, vis, t(case["campos"]), t(case["level"]), t(case["opacity_scale"]), mo)
    leaves.update(par)
    return out, leaves


def backward(out, leaves, dL, device="cpu"):
    loss = sum((out[n] * torch.tensor(dL[n], dtype=out[n].dtype, device=device)).sum() for n in ("xyz", "color", "opacity", "scaling", "rot"))
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    return {n: (None if g is None else g.cpu().numpy()) for n, g in zip(names, grads)}


def training_statis(acc, n_offsets, viewspace_grad, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    """Torch restatement of ScaffoldGaussian.training_statis (gssr/gaussian/scaffold_gaussian.py:488-508) for timing the op chain;
    acc: dict of the four accumulators, updated in place."""
    temp = opacity.clone().view(-1).detach()
    temp[temp < 0] = 0
    temp = temp.view([-1, n_offsets])
    acc["opacity_accum"][anchor_visible_mask] += temp.sum(dim=1, keepdim=True)
    acc["anchor_demon"][anchor_visible_mask] += 1
    avm = anchor_visible_mask.unsqueeze(dim=1).repeat([1, n_offsets]).view(-1)
    combined = torch.zeros_like(acc["offset_gradient_accum"], dtype=torch.bool).squeeze(dim=1)
    combined[avm] = offset_selection_mask
    tmp = combined.clone()
    combined[tmp] = update_filter
    grad_norm = torch.norm(viewspace_grad[update_filter, :2], dim=-1, keepdim=True)
    acc["offset_gradient_accum"][combined] += grad_norm
    acc["offset_denom"][combined] += 1
