"""Worker side of tests/test_gpu_multi.py (run under torch.distributed.run with the nccl backend = RCCL, one rank per GPU), plus the
per-tile entry points the launcher test uses."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gs-sr_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def tile_entry(tile_dir, out_paths, device, tile_index):
    """Records how the launcher pinned this worker; 'trains' 7 iterations."""
    json.dump({"tile": tile_index, "visible": os.environ.get("HIP_VISIBLE_DEVICES"), "device_count": torch.cuda.device_count(),
               "device": str(device), "pid": os.getpid(), "affinity": len(os.sched_getaffinity(0))}, open(os.path.join(out_paths["config"], "worker.json"), "w"))
    x = torch.ones(8, device=device)
    assert float(x.sum()) == 8.0
    return 7


def failing_entry(tile_dir, out_paths, device, tile_index):
    if tile_index == 0:
        raise SystemExit(3)
    return 1


def main():
    import torch.distributed as dist
    import numpy as np
    from gsrast import tiles
    from gsrast.tsdf import ScalableTSDFVolume
    import tsdf_cases
    import oracle
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # 1. optional gradient all-reduce on shared anchor rows (north_star; off by default in the product)
    g = torch.Generator().manual_seed(rank)
    grads = [torch.randn(10, 3, generator=g).to(dev), torch.randn(4, 4, generator=g).to(dev)]
    rows = [torch.tensor([1, 3, 5], device=dev), None]
    ref = [torch.stack([torch.randn(10, 3, generator=torch.Generator().manual_seed(r)) for r in range(world)]),
           torch.stack([torch.randn(10, 3, generator=torch.Generator().manual_seed(r)) for r in range(world)])]
    before = grads[0].clone()
    n = tiles.allreduce_shared_(grads, rows, average=True)
    assert n == 3 * 3 + 16
    assert torch.allclose(grads[0][[1, 3, 5]].cpu(), ref[0].mean(0)[[1, 3, 5]], atol=1e-6)
    assert torch.equal(grads[0][[0, 2, 4]], before[[0, 2, 4]])
    # 2. multi-tile TSDF fusion: each rank integrates its own frames, merge_() fuses; compare with the oracle's joint integration
    frs = tsdf_cases.frames(2 * world, seed=5)
    vol = ScalableTSDFVolume(0.02, 0.1, capacity_units=4096, device=dev)
    for f in frs[rank::world]:
        vol.integrate(torch.from_numpy(f["rgb"]).to(dev), torch.from_numpy(f["depth"]).to(dev), f["fx"], f["fy"], f["cx"], f["cy"], f["E"], depth_trunc=6.0)
    vol.merge_()
    o = oracle.SparseTSDF(0.02, 0.1)
    for f in frs:
        o.integrate(tsdf_cases.rgb8(f["rgb"]), f["depth"], f["fx"], f["fy"], f["cx"], f["cy"], f["E"], depth_trunc=6.0)
    rco, rt, rw, rc = o.units()
    co, t, w, c = (x.cpu().numpy() for x in vol.units())
    want = {tuple(k): i for i, k in enumerate(rco.tolist())}
    assert set(map(tuple, co.tolist())) == set(want)
    bad = tot = 0
    for i, k in enumerate(co.tolist()):
        j = want[tuple(k)]
        bad += int((w[i] != rw[j]).sum()) + int((np.abs(t[i] - rt[j])[w[i] == rw[j]] > 2e-4).sum()); tot += w[i].size
    assert bad <= 2e-4 * tot, (bad, tot)
    # 3. barrier + job reduction on device tensors (bench.py's timing path)
    tiles.barrier(dev)
    tmax, nsum = tiles.reduce_job(1.0 + rank, 10, dev)
    assert tmax == float(world) and nsum == 10 * world
    dist.barrier()
    if rank == 0:
        print("multi_rank_worker ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
