"""Fused Adam for the Gaussian models (include/gsrast.h gsr_adam_step).

The reference builds `torch.optim.Adam(l, lr=0.0, eps=1e-15)` with one parameter group per tensor (gssr/gaussian/vanilla_gaussian.py:120-139,
scaffold_gaussian.py setup_optimizers), rewrites `param_group['lr']` from its schedulers, steps it once per iteration
(gssr/engine/trainer.py:127-128) and edits `optimizer.state[param]['exp_avg' / 'exp_avg_sq']` when it densifies or prunes
(`cat_tensors_to_optimizer`, `_prune_optimizer`, `replace_tensor_to_optimizer`).  `gsrast.optim.Adam` is a subclass of `torch.optim.Adam`
with the same constructor, param groups, state keys ('step', 'exp_avg', 'exp_avg_sq') and state_dict, so all of that code runs unchanged;
only `step()` differs: every float32 HIP parameter is updated by ONE streaming kernel (28 bytes per parameter) instead of torch's
op-by-op update.  Parameters the kernel does not cover (other dtypes / devices, sparse gradients) and the options it does not implement
(amsgrad, weight_decay, maximize, capturable, differentiable) go through torch's own implementation.
A param group may carry `lr_scale` (float32 tensor shaped like its parameter): a per-element multiplier of the group's learning rate, for
models that keep all their parameters in one flat tensor (bench.py).

HIP graphs (round 3): a `step()` issued while the current stream is being captured (torch.cuda.graph) records ONE launch whose per-step scalars
(step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t)) are read from a small device buffer instead of being baked into the kernel arguments
(gsr_adam_step_multi_dev).  `prepare_replay()` -- call it before every `graph.replay()`; gsrast.graphs.GraphedStep does -- bumps the step counters,
re-reads every group's `lr` (schedulers keep working) and refreshes that buffer with one small asynchronous copy.

Two render passes over the same parameters (PGSR's reference + neighbour camera, gssr/scene/pgsr_scene.py:226-338): autograd sums the two passes'
gradients with one `add` kernel per parameter tensor (27 launches, 130 us per octree-pgsr iteration).  `shadow_parameters(x)` returns a second set of
leaves over the SAME storage for the second pass; `Adam.add_shadows(params, shadows)` makes step() read both gradients in the update kernel
(grad + grad2, the same single fp32 add) and zero_grad() clear both."""
import ctypes as C
import math

import torch
from torch.optim.adam import adam as _torch_adam

from . import check, lib, stream_ptr


class _AdamTensor(C.Structure):          # include/gsrast.h gsr_adam_tensor
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("lr_scale", C.c_void_p),
                ("n", C.c_int64), ("beta1", C.c_double), ("beta2", C.c_double), ("step_size", C.c_float), ("bias_correction2_sqrt", C.c_float),
                ("eps", C.c_float), ("pad_", C.c_float), ("grad2", C.c_void_p)]


def shadow_parameters(x):
    """A second set of autograd leaves over the SAME storage as `x` (a tensor, a list / tuple / dict of them, or an nn.Module, whose structure is
    copied): what the second of two render passes of one iteration reads, so that its gradients arrive in their own .grad instead of being added
    to the first pass's by one kernel per tensor.  In-place updates of the originals (the optimizer's) are seen by the shadows."""
    if isinstance(x, torch.nn.Module):
        import copy
        memo = {id(q): torch.nn.Parameter(q.detach(), requires_grad=q.requires_grad) for q in x.parameters()}
        memo.update({id(b): b for b in x.buffers()})               # buffers are shared as they are
        return copy.deepcopy(x, memo)
    if isinstance(x, torch.Tensor):
        return x.detach().requires_grad_(x.requires_grad)
    if isinstance(x, dict):
        return {k: shadow_parameters(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(shadow_parameters(v) for v in x)
    raise TypeError(f"shadow_parameters: {type(x)}")


def _leaves(x):
    if isinstance(x, torch.nn.Module):
        return list(x.parameters())
    if isinstance(x, torch.Tensor):
        return [x]
    if isinstance(x, dict):
        return [t for v in x.values() for t in _leaves(v)]
    return [t for v in x for t in _leaves(v)]


def _covered(p, group):
    return (p.is_cuda and p.dtype == torch.float32 and p.grad is not None and not p.grad.is_sparse and p.grad.dtype == torch.float32
            and p.is_contiguous() and not group.get("amsgrad", False) and group.get("weight_decay", 0) == 0 and not group.get("maximize", False)
            and not group.get("capturable", False) and not group.get("differentiable", False))


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.setdefault("foreach", False); kw.setdefault("fused", False)        # the fallback path: torch's plain per-tensor implementation
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)

    def add_shadows(self, params, shadows):
        """`shadows` = shadow_parameters(`params`) (same structure): step() adds a shadow's gradient to its parameter's inside the update
        kernel, zero_grad() clears it.  One shadow per parameter."""
        sh = self.__dict__.setdefault("_gsr_shadows", {})
        ps, ss = _leaves(params), _leaves(shadows)
        if len(ps) != len(ss):
            raise ValueError("add_shadows: params and shadows differ in structure")
        for p, q in zip(ps, ss):
            if q.data_ptr() != p.data_ptr() or q.shape != p.shape:
                raise ValueError("add_shadows: a shadow must share its parameter's storage (shadow_parameters)")
            sh[id(p)] = q

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)
        for q in self.__dict__.get("_gsr_shadows", {}).values():
            if q.grad is not None:
                if set_to_none:
                    q.grad = None
                else:
                    q.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        shadows = self.__dict__.get("_gsr_shadows", {})
        L = lib()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        captured = []                              # capture: (param, group) in table order, for prepare_replay
        steps = self.__dict__.setdefault("_gsr_step_views", {})
        if len(steps) > 4 * sum(len(g["params"]) for g in self.param_groups) + 64:      # parameters replaced by densification leave stale ids
            steps.clear()
        rest = []                                  # (group, params) torch handles itself
        batch = {}                                 # device -> list of table entries (one launch per 24 tensors)
        keep = []                                  # contiguous copies that must outlive the launch call
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr = group["lr"]
            if isinstance(lr, torch.Tensor):
                lr = float(lr)
            sc = group.get("lr_scale")             # optional: one learning-rate multiplier per element (float32, shaped like the parameter)
            other = []
            for p in group["params"]:
                g2 = None
                if shadows:
                    q = shadows.get(id(p))
                    if q is not None and q.grad is not None:
                        if p.grad is None:
                            p.grad = q.grad; q.grad = None         # only the second pass reached this parameter
                        else:
                            g2 = q.grad
                if p.grad is None:
                    continue
                cov = _covered(p, group)
                if g2 is not None and not (cov and g2.dtype == torch.float32 and not g2.is_sparse and g2.shape == p.grad.shape):
                    p.grad = p.grad + g2; g2 = None                # the torch path below knows one gradient
                if not cov:
                    other.append(p)
                    continue
                st = self.state[p]
                if len(st) == 0:                   # same lazy state as torch.optim.Adam._init_group
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                stp = st["step"]                   # a CPU scalar tensor, as torch keeps it; bumped through a cached numpy view of the same
                ent = steps.get(id(p))             # memory (a torch op + float() per parameter cost more than the whole kernel launch)
                if ent is None or ent[0] is not stp:
                    ent = steps[id(p)] = (stp, stp.numpy()) if (stp.device.type == "cpu" and stp.dtype == torch.float32) else (stp, None)
                view = ent[1]
                if capturing:                      # nothing runs during capture: the counter moves in prepare_replay()
                    t = max(float(stp), 1.0)
                    captured.append((p, group))
                elif view is not None:
                    view += 1.0
                    t = float(view)
                else:
                    stp += 1
                    t = float(stp)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    st["exp_avg"], st["exp_avg_sq"] = m, v = m.contiguous(), v.contiguous()
                g = p.grad
                if not g.is_contiguous():
                    g = g.contiguous(); keep.append(g)
                if g2 is not None and not g2.is_contiguous():
                    g2 = g2.contiguous(); keep.append(g2)
                if sc is not None and (sc.numel() != p.numel() or sc.dtype != torch.float32 or not sc.is_contiguous() or sc.device != p.device):
                    raise RuntimeError("gsrast.optim.Adam: lr_scale must be a contiguous float32 tensor with one entry per parameter element")
                batch.setdefault(p.device, []).append(
                    (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 0 if sc is None else sc.data_ptr(), p.numel(), beta1, beta2,
                     lr / (1.0 - beta1 ** t), math.sqrt(1.0 - beta2 ** t), group["eps"], 0.0, 0 if g2 is None else g2.data_ptr()))
            if other:
                rest.append((group, other))
        if capturing:
            if rest or len(batch) > 1:
                raise RuntimeError("gsrast.optim.Adam: a captured step() needs every parameter covered by the HIP kernel, on one device")
            graphs = self.__dict__.setdefault("_gsr_graph", [])       # one entry per captured step(), kept for the life of the optimizer: several
                                                                       # graphs over one optimizer (one per image size, say) each replay their own
        for dev, entries in batch.items():
            table = (_AdamTensor * len(entries))(*entries)
            with torch.cuda.device(dev):
                if capturing:
                    # The buffer the recorded launch reads its per-step scalars from must NOT come from the graph's private pool: a block of that
                    # pool may have held a temporary earlier in the recorded iteration, whose recorded writes would then overwrite, on every
                    # replay, what prepare_replay() copied in beforehand (measured: step_size read back as 0).  It is created by the eager
                    # step() calls that precede every capture (torch.cuda.graph needs warm-up iterations anyway).
                    # Every capture takes the spare buffer for itself (its recorded launch reads table position i of it for the i-th captured tensor);
                    # the next eager step() provides a new spare.
                    hyper = self.__dict__.get("_gsr_hyper", {}).pop(dev, None)
                    if hyper is None or hyper.shape[0] < len(entries):
                        raise RuntimeError("gsrast.optim.Adam: run at least one eager step() before recording each step() into a graph")
                    graphs.append([captured, hyper, [], 0])
                    check(L.gsr_adam_step_multi_dev(len(entries), table, hyper.data_ptr(), stream_ptr(dev)), "adam_step_multi_dev")
                else:
                    hy = self.__dict__.setdefault("_gsr_hyper", {})
                    if dev not in hy or hy[dev].shape[0] < len(entries):
                        hy[dev] = torch.zeros(len(entries) + 8, 2, dtype=torch.float32, device=dev)
                    check(L.gsr_adam_step_multi(len(entries), table, stream_ptr(dev)), "adam_step_multi")
        if capturing:
            return loss
        for group, ps in rest:                     # uncovered parameters: exactly torch's update (its functional form on a copy of the group
            g2 = dict(group, params=ps)            # that holds only them) -- not super().step(), which would fire the step hooks a second time
            pw, grads, m1, m2, mx, st = [], [], [], [], [], []
            has_complex = self._init_group(g2, pw, grads, m1, m2, mx, st)
            b1, b2 = group["betas"]
            _torch_adam(pw, grads, m1, m2, mx, st, amsgrad=group["amsgrad"], has_complex=has_complex, beta1=b1, beta2=b2, lr=group["lr"],
                        weight_decay=group["weight_decay"], eps=group["eps"], maximize=group["maximize"], foreach=group["foreach"],
                        capturable=group["capturable"], differentiable=group["differentiable"], fused=group["fused"],
                        grad_scale=getattr(self, "grad_scale", None), found_inf=getattr(self, "found_inf", None),
                        decoupled_weight_decay=group.get("decoupled_weight_decay", False))
        return loss

    def captured_steps(self):
        """Handles of the step() calls recorded into graphs so far (in capture order): `opt.captured_steps()[n0:]` after a capture that started with
        n0 = len(opt.captured_steps()) are that capture's; pass them to prepare_replay(handles=...)."""
        return list(self.__dict__.get("_gsr_graph", ()))

    @torch.no_grad()
    def prepare_replay(self, handles=None):
        """Before every replay of a graph that holds this optimizer's step(): one more step for every captured parameter -- counters bumped,
        the groups' current learning rates and the bias corrections written to the device buffer the captured launch reads (one small
        asynchronous copy from a ring of pinned host buffers: the host may run several replays ahead of the device).  `handles`: the captured
        steps of THE GRAPH ABOUT TO BE REPLAYED (captured_steps(); gsrast.graphs.GraphedStep passes its own); None = every captured step, which is
        right only while one graph holds this optimizer."""
        for ent in (self.__dict__.get("_gsr_graph", ()) if handles is None else handles):
            captured, hyper, ring, pos = ent
            if len(ring) < 16:
                ring.append([torch.empty(len(captured), 2, dtype=torch.float32).pin_memory(), None])
                slot = ring[-1]
            else:
                slot = ring[pos % 16]
                slot[1].synchronize()              # the copy that last used this pinned buffer has completed (16 replays ago: normally no wait)
            ent[3] = pos + 1
            h = slot[0].numpy()
            for i, (p, group) in enumerate(captured):
                st = self.state[p]
                st["step"] += 1
                t = float(st["step"])
                beta1, beta2 = group["betas"]
                h[i, 0] = float(group["lr"]) / (1.0 - beta1 ** t)
                h[i, 1] = math.sqrt(1.0 - beta2 ** t)
            hyper[: len(captured)].copy_(slot[0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            slot[1] = ev
