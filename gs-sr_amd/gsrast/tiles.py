"""One-tile-per-GPU sharding of VastGaussian-partitioned scenes (replaces the sequential loop of train_split.py:15-38).

Tiles are self-contained sub-scenes (own images, points, anchors, optimiser; split_scene.py:55-82): they shard with NO
data-path collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the start/stop barrier
and the max-over-ranks reduction of elapsed time / aggregation of per-tile iteration counts.
"""
import os
import re

import torch
import torch.distributed as dist


def list_tiles(data_dir):
    """tile_<digits> sub-directories in SORTED order.  The reference (train_split.py:15-16) takes os.listdir() order filtered on
    startswith('tile_') -- unsorted and platform-dependent -- so its tile_%04d OUTPUT index follows listdir order; here the index
    follows the sorted names (deterministic across ranks, which the sharding needs).  INTEGRATION.md notes the difference for anyone
    comparing per-tile outputs with a reference run."""
    return sorted(d for d in os.listdir(data_dir) if re.fullmatch(r"tile_\d+", d) and os.path.isdir(os.path.join(data_dir, d)))


def assign_tiles(num_tiles, world_size, rank):
    """Round-robin: rank r owns tiles r, r+world, ...  (4 tiles on 4 GPUs / 8 on 8 -> exactly one tile per GPU)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, num_tiles, world_size))


def tile_output_paths(base_output_dir, tile_name):
    """Per-tile output sub-paths exactly as train_split.py:28-35 rewrites them."""
    return {k: os.path.join(base_output_dir, tile_name, k) for k in ("chkpnt", "point_cloud", "tb", "config")}


def barrier(device=None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def reduce_job(elapsed_s, iters_done, device=None):
    """-> (max elapsed over ranks, total iterations over ranks): the whole-job throughput is total / max."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), int(iters_done)
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    n = torch.tensor([int(iters_done)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def allreduce_shared_(grads, shared_rows=None, average=True):
    """OPTIONAL build extension (no reference counterpart: the reference never synchronises tiles) -- OFF unless called.

    Sums (or averages) the gradients of parameters that several tile workers share -- the overlap-band anchors of neighbouring
    VastGaussian tiles (`shared_rows`: one LongTensor of local row indices per tensor, same length and order on every rank) or whole
    tensors such as the three decode MLPs / the appearance embedding (`shared_rows[i] is None`).  Everything is packed into ONE flat fp32
    bucket and reduced with a single all-reduce: over xGMI the payload (a few MB) is latency-bound, so one collective per iteration is the
    right shape (RCCL picks its direct/tree path for small messages; a per-tensor loop would pay the launch latency a dozen times).
    In place; returns the number of floats communicated."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    if shared_rows is None:
        shared_rows = [None] * len(grads)
    parts = [(g if r is None else g.index_select(0, r)).reshape(-1).float() for g, r in zip(grads, shared_rows)]
    flat = torch.cat(parts)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g, r, p in zip(grads, shared_rows, parts):
        n = p.numel()
        red = flat[off:off + n].to(g.dtype)
        if r is None:
            g.copy_(red.view_as(g))
        else:
            g.index_copy_(0, r, red.view(r.numel(), *g.shape[1:]))
        off += n
    return int(flat.numel())
