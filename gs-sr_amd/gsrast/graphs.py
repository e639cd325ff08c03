"""One training iteration as a HIP graph (round 3, VERDICT r2 #4b).

The reference's iteration (gssr/engine/trainer.py:84-140: render -> loss -> backward -> densify bookkeeping -> optimizer.step) is ~80-190 kernel
launches issued from Python; at 300k Gaussians / 1080p the GPU needs ~1.6 ms for them and the host ~1.9 ms to issue them.  Between two
densification steps every shape is static, so the whole iteration can be recorded once and replayed:

    it = GraphedStep(step_fn, optimizers=[opt])      # step_fn(): forward, loss, backward, statistics, opt.step(), zero_grad(set_to_none=True)
    for _ in range(n): it()                          # replays; camera / target tensors are updated IN PLACE by the caller between calls
    it.check()                                       # synchronises: raises if a rasterizer arena overflowed since the last check

What makes the iteration recordable (all opt-in, the default paths keep their exact reference semantics):
  * rasterizer forward without its host sync (gsr_forward_async): fixed binning capacity = `headroom` x the instance count of the eager warm-up
    runs, device-side overflow flag (gsrast.rasterize.async_status);
  * neural-Gaussian decode with static_rows=True (gsd_forward_static): all Nv*k rows, the unused ones parked where every rasterizer culls them;
  * gsrast.optim.Adam.step() under capture reads its per-step scalars from a device buffer that prepare_replay() refreshes.
Anything that changes a shape (densification, pruning, another image size) needs a new GraphedStep."""
import torch

from . import rasterize


class GraphedStep:
    def __init__(self, fn, optimizers=(), warmup=3, check_every=0):
        self.fn, self.opts, self.check_every, self.calls = fn, list(optimizers), int(check_every), 0
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):              # eager warm-up on a side stream, as torch.cuda.graph asks: creates optimizer state, capacity hints
            for _ in range(max(1, warmup)):
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        rasterize.async_status_reset()
        n0 = [len(o.captured_steps()) if hasattr(o, "captured_steps") else 0 for o in self.opts]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()
        # the optimizer steps recorded into THIS graph: only they move when it is replayed (another GraphedStep over the same optimizer has its own)
        self.opt_steps = [o.captured_steps()[k:] if hasattr(o, "captured_steps") else None for o, k in zip(self.opts, n0)]
        self.status = list(rasterize._ASYNC_STATUS)          # the sync-free forwards recorded into this graph
        rasterize.async_status_reset()

    def __call__(self):
        for o, h in zip(self.opts, self.opt_steps):
            if h is None:
                o.prepare_replay()
            else:
                o.prepare_replay(handles=h)
        self.graph.replay()
        self.calls += 1
        if self.check_every and self.calls % self.check_every == 0:
            self.check()
        return self.out

    def check(self):
        """Synchronises and raises when a recorded rasterizer forward ran out of binning capacity since the last check (the outputs of those
        replays were incomplete): re-create the GraphedStep -- its warm-up re-measures the instance count."""
        res = []
        for st, cap in self.status:
            h = st.cpu()
            res.append((int(h[0]), bool(h[1]), cap))
            if bool(h[2]):
                st.zero_()
                raise RuntimeError(rasterize.PREFILTERED_MSG)
            if bool(h[1]):
                st.zero_()
                raise RuntimeError(f"gsrast.graphs: a recorded rasterizer forward overflowed its binning arena ({int(h[0])} instances, capacity {cap})")
        return res
