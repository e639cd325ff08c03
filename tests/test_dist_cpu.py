"""world_size=2 gloo test (CPU) of the one-tile-per-GPU sharding path used by bench.py --gpus N / the tile launcher."""
import os
import socket
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, tmp):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gs-sr_amd"))
    from gsrast import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names = tiles.list_tiles(tmp)
    mine = tiles.assign_tiles(len(names), world, rank)
    tiles.barrier()
    # every tile owned exactly once across ranks
    owned = torch.zeros(len(names), dtype=torch.int64)
    owned[mine] = 1
    dist.all_reduce(owned)
    assert bool((owned == 1).all())
    elapsed, total = tiles.reduce_job(1.0 + rank, 10 * len(mine))
    assert elapsed == float(world) and total == 10 * len(names)
    paths = tiles.tile_output_paths("/out", names[mine[0]])
    assert paths["chkpnt"] == f"/out/{names[mine[0]]}/chkpnt"
    dist.barrier()
    dist.destroy_process_group()


def test_tile_sharding_world2_gloo():
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(5):
            os.makedirs(os.path.join(tmp, f"tile_{i:04d}"))
        os.makedirs(os.path.join(tmp, "not_a_tile"))
        mp.spawn(_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)


def test_assign_tiles_covers_configs():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gs-sr_amd"))
    from gsrast import tiles
    assert [tiles.assign_tiles(4, 4, r) for r in range(4)] == [[0], [1], [2], [3]]          # BASELINE config 4
    assert [tiles.assign_tiles(8, 8, r) for r in range(8)] == [[r] for r in range(8)]       # BASELINE config 5
    assert sorted(sum((tiles.assign_tiles(8, 3, r) for r in range(3)), [])) == list(range(8))
