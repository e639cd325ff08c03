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


def _merge_worker(rank, world, port):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gs-sr_amd"))
    from gsrast.tsdf import merge_volumes_
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    # two "frames" integrated sequentially on one rank vs one frame per rank + merge: same weighted average
    obs = torch.rand(2, 4, 5, 6, generator=g); seen = torch.rand(2, 4, 5, 6, generator=g) > 0.3
    col = torch.rand(2, 4, 5, 6, 3, generator=g)
    tsdf = torch.where(seen[rank], obs[rank], torch.zeros(4, 5, 6)); w = seen[rank].float()
    c = torch.where(seen[rank].unsqueeze(-1), col[rank], torch.zeros(4, 5, 6, 3))
    merge_volumes_(tsdf, w, c)
    wt = seen.float().sum(0)
    exp = torch.where(wt > 0, (obs * seen).sum(0) / wt.clamp_min(1), torch.zeros(4, 5, 6))
    assert torch.allclose(w, wt) and torch.allclose(tsdf, exp, atol=1e-6)
    assert torch.allclose(c, torch.where((wt > 0).unsqueeze(-1), (col * seen.unsqueeze(-1)).sum(0) / wt.clamp_min(1).unsqueeze(-1),
                                         torch.zeros(4, 5, 6, 3)), atol=1e-6)
    dist.barrier(); dist.destroy_process_group()


def test_tsdf_volume_merge_world2_gloo():
    mp.spawn(_merge_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_assign_tiles_covers_configs():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gs-sr_amd"))
    from gsrast import tiles
    assert [tiles.assign_tiles(4, 4, r) for r in range(4)] == [[0], [1], [2], [3]]          # BASELINE config 4
    assert [tiles.assign_tiles(8, 8, r) for r in range(8)] == [[r] for r in range(8)]       # BASELINE config 5
    assert sorted(sum((tiles.assign_tiles(8, 3, r) for r in range(3)), [])) == list(range(8))


def test_launch_tiles_two_workers_gloo():
    """gsrast.launch_tiles: 3 tiles over 2 worker processes (gloo): every tile trained once, output dirs as train_split.py:28-35."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "scene"); out = os.path.join(tmp, "out")
        for i in (3, 1, 7):
            os.makedirs(os.path.join(data, f"tile_{i:04d}"))
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "gs-sr_amd"), os.path.join(root, "tests"),
                                                            os.environ.get("PYTHONPATH", "")]))
        env.pop("RANK", None); env.pop("WORLD_SIZE", None)
        r = subprocess.run([sys.executable, "-m", "gsrast.launch_tiles", "--data", data, "--output", out, "--gpus", "2",
                            "--backend", "gloo", "--port", str(_free_port()), "--entry", "tile_entry_fixture:train_tile"],
                           env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        summary = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert summary["tiles"] == 3 and summary["workers"] == 2 and summary["iterations"] == 21
        ranks = set()
        for k, src in enumerate(("tile_0001", "tile_0003", "tile_0007")):      # sorted source tiles -> tile_%04d index dirs
            for sub in ("chkpnt", "point_cloud", "tb", "config"):
                assert os.path.isdir(os.path.join(out, f"tile_{k:04d}", sub))
            txt = open(os.path.join(out, f"tile_{k:04d}", "chkpnt", "done.txt")).read().split()
            assert txt[0] == src and int(txt[1]) == k and txt[2] == "cpu"
            ranks.add(txt[3])
        assert ranks == {"rank0", "rank1"}
        # a rank that dies must end the job promptly (the surviving rank would otherwise sit in the barrier until the backend times out)
        import time
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "gsrast.launch_tiles", "--data", data, "--output", out + "2", "--gpus", "2",
                            "--backend", "gloo", "--port", str(_free_port()), "--entry", "tile_entry_fixture:train_tile_rank1_dies"],
                           env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode != 0 and time.time() - t0 < 120


def _shared_worker(rank, world, port):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gs-sr_amd"))
    from gsrast import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    feat = torch.randn(50, 32, generator=g); W = torch.randn(32, 35, generator=g); b = torch.randn(32, generator=g)
    keep = [feat.clone(), W.clone(), b.clone()]
    # each rank shares 10 anchor rows (different local indices, same logical anchors, same order) + the whole MLP layer
    rows = torch.arange(5, 15) if rank == 0 else torch.arange(30, 40)
    n = tiles.allreduce_shared_([feat, W, b], [rows, None, None], average=True)
    assert n == 10 * 32 + 32 * 35 + 32
    other = torch.Generator().manual_seed(100 + (1 - rank))
    ofeat = torch.randn(50, 32, generator=other); oW = torch.randn(32, 35, generator=other); ob = torch.randn(32, generator=other)
    orows = torch.arange(30, 40) if rank == 0 else torch.arange(5, 15)
    assert torch.allclose(feat[rows], 0.5 * (keep[0][rows] + ofeat[orows]))
    mask = torch.ones(50, dtype=torch.bool); mask[rows] = False
    assert torch.equal(feat[mask], keep[0][mask])                      # private rows untouched
    assert torch.allclose(W, 0.5 * (keep[1] + oW)) and torch.allclose(b, 0.5 * (keep[2] + ob))
    dist.barrier(); dist.destroy_process_group()


def test_allreduce_shared_anchor_grads_world2_gloo():
    mp.spawn(_shared_worker, args=(2, _free_port()), nprocs=2, join=True)


def _sparse_merge_worker(rank, world, port):
    """Every rank owns a different set of block-sparse TSDF units (with overlap); gather_unit_lists + merge_unit_lists -- the transport of
    ScalableTSDFVolume.merge_() -- must give every rank the same weighted fusion."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gs-sr_amd"))
    from gsrast.tsdf import gather_unit_lists, merge_unit_lists
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(11)
    V = 64                                                    # voxels per unit in this test (the helpers are shape-agnostic)
    all_co = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 2, -1], [5, 5, 5], [-3, 1, 0]], dtype=torch.int32)
    own = [[0, 1, 2], [1, 2, 3, 4]][rank]                      # units 1 and 2 are seen by both ranks
    t_all = torch.rand(2, 5, V, generator=g); w_all = torch.randint(0, 4, (2, 5, V), generator=g).float(); c_all = torch.rand(2, 5, V, 3, generator=g)
    co, t, w, c = gather_unit_lists(all_co[own], t_all[rank][own], w_all[rank][own], c_all[rank][own])
    assert co.shape[0] == 7
    mco, mt, mw, mc = merge_unit_lists(co, t, w, c)
    assert mco.shape[0] == 5
    for i, k in enumerate(mco.tolist()):
        j = [tuple(x) for x in all_co.tolist()].index(tuple(k))
        ws = [w_all[r][j] for r in range(2) if j in [[0, 1, 2], [1, 2, 3, 4]][r]]
        ts = [t_all[r][j] for r in range(2) if j in [[0, 1, 2], [1, 2, 3, 4]][r]]
        wt = sum(ws)
        exp = torch.where(wt > 0, sum(a * b for a, b in zip(ts, ws)) / wt.clamp_min(1e-30), torch.zeros(V))
        assert torch.equal(mw[i], wt) and torch.allclose(mt[i], exp, atol=1e-6)
    dist.barrier(); dist.destroy_process_group()


def test_sparse_tsdf_merge_world2_gloo():
    mp.spawn(_sparse_merge_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_launch_tiles_two_workers_per_gpu_gloo():
    """--workers-per-gpu 2 on 2 'GPUs' (gloo dry run): four ranks, every one of 5 tiles trained exactly once, rank r on device r % 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "scene"); out = os.path.join(tmp, "out")
        for i in range(5):
            os.makedirs(os.path.join(data, f"tile_{i:04d}"))
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "gs-sr_amd"), os.path.join(root, "tests"), os.environ.get("PYTHONPATH", "")]))
        env.pop("RANK", None); env.pop("WORLD_SIZE", None)
        r = subprocess.run([sys.executable, "-m", "gsrast.launch_tiles", "--data", data, "--output", out, "--gpus", "2", "--workers-per-gpu", "2",
                            "--backend", "gloo", "--port", str(_free_port()), "--entry", "tile_entry_fixture:train_tile"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        summary = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert summary["tiles"] == 5 and summary["workers"] == 4 and summary["iterations"] == 35
        ranks = [open(os.path.join(out, f"tile_{k:04d}", "chkpnt", "done.txt")).read().split()[3] for k in range(5)]
        assert set(ranks) == {"rank0", "rank1", "rank2", "rank3"}


def test_rank_placement_follows_the_visible_device_list_by_pci_address(tmp_path):
    """spawn_ranks pins rank r to the (r % gpus)-th entry of the launcher's HIP_VISIBLE_DEVICES and to the cores of THAT device's NUMA node, resolved by
    PCI address from the KFD topology -- not by the rank's ordinal or the order of /sys/class/drm/card* (VERDICT r3 weak #7).  A fake sysfs root: an
    8-GPU node whose KFD order differs from its PCI order, two NUMA nodes, HIP_VISIBLE_DEVICES=4,5,6,7 and a device named by its unique id."""
    from gsrast import launch_tiles
    root = tmp_path
    buses = [0x0c, 0x22, 0x38, 0x5c, 0x9f, 0xaf, 0xbf, 0xdf]         # KFD GPU order -> PCI bus
    nodes = root / "sys/class/kfd/kfd/topology/nodes"
    for cpu in (0, 1):                                                 # two CPU nodes first, as on a real 2-socket box
        d = nodes / str(cpu); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\nunique_id 0\n")
    for k, bus in enumerate(buses):
        d = nodes / str(2 + k); d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\nunique_id {0xabc000 + k}\n")
        pd = root / "sys/bus/pci/devices" / f"0000:{bus:02x}:00.0"; pd.mkdir(parents=True)
        (pd / "numa_node").write_text(f"{0 if k < 4 else 1}\n")
    for node, cl in ((0, "0-63,128-191"), (1, "64-127,192-255")):
        nd = root / f"sys/devices/system/node/node{node}"; nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    assert [b for b, _ in launch_tiles.hip_device_bdfs(str(root))] == [f"0000:{b:02x}:00.0" for b in buses]
    place = launch_tiles.rank_placement(4, 4, "4,5,6,7", sysfs_root=str(root))
    assert [d for d, _ in place] == ["4", "5", "6", "7"]
    assert all(c is not None and min(c) == 64 and max(c) == 255 and 0 not in c for _, c in place)           # GPUs 4..7 sit on node 1
    place = launch_tiles.rank_placement(8, 8, None, sysfs_root=str(root))
    assert [min(c) for _, c in place] == [0, 0, 0, 0, 64, 64, 64, 64]
    place = launch_tiles.rank_placement(2, 2, f"GPU-{0xabc006:x},1", sysfs_root=str(root))
    assert min(place[0][1]) == 64 and min(place[1][1]) == 0                                                 # by unique id, then by index
    place = launch_tiles.rank_placement(4, 2, "6,1", sysfs_root=str(root))                                  # two workers per GPU share a placement
    assert [d for d, _ in place] == ["6", "1", "6", "1"]
    assert launch_tiles.gpu_numa_cpus("11", str(root)) is None and launch_tiles.gpu_numa_cpus("GPU-dead", str(root)) is None
    assert all(c is None for _, c in launch_tiles.rank_placement(2, 2, "0,1", sysfs_root=str(root), affinity=False))
