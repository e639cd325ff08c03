"""Tile-parallel launcher: one worker process per GPU, one VastGaussian tile at a time per worker.

Replaces the sequential loop of the reference (train_split.py:24-38: deep-copy the config per tile, rewrite the four relative
output dirs to tile_%04d/<dir>, call train.main) by N concurrent workers.  Tiles are independent sub-scenes, so there is no data-path
collective; torch.distributed (RCCL on GPUs / gloo on CPU) carries only the start/stop barrier and the (max elapsed, sum iterations)
reduction that yields the whole-job it/s.

    python -m gsrast.launch_tiles --data <source_path> --output <dir> --gpus 4 --entry mypkg.train:train_tile [--backend nccl]
                                  [--workers-per-gpu 2]

`--workers-per-gpu K` starts K workers on every GPU (K * gpus ranks, rank r on GPU r % gpus).  One training iteration at 300k Gaussians /
1080p leaves the GPU idle between its ~90 dependent launches; two independent tiles interleave on one MI355X at 1081 it/s in total against
904 it/s for one (profiles/r02_bench_2ranks_one_gpu.json; three or four workers lose again: 823 / 920 it/s), and 288 GB hold many tiles.
RCCL refuses a communicator in which two ranks sit on the same device ("Duplicate GPU detected"), and the job has no data-path collective
anyway, so with K > 1 the workers stay on their HIP devices while the control collectives (start/stop barrier, job reduction) run over gloo
on host tensors: `--backend` names the CONTROL backend, `--device` where the tiles train (default: cuda for nccl, cpu for gloo).

`--entry module:function` names the per-tile trainer: `function(tile_dir, out_paths, device, tile_index) -> iterations_done`.
For the reference that function is a 5-line shim around `train.main(tile_config)`.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

from . import tiles


def resolve_entry(spec):
    mod, _, fn = spec.partition(":")
    if not mod or not fn:
        raise ValueError("--entry must be module:function")
    return getattr(importlib.import_module(mod), fn)


def tile_dirnames(num_tiles):
    """tile_%04d, the naming train_split.py:17 gives config_of_tiles (index order, not directory-name order)."""
    return ["tile_%04d" % i for i in range(num_tiles)]


def worker(args):
    """Body of one rank.  Returns the job summary dict on rank 0, None elsewhere."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    use_gpu = worker_device(args) == "cuda"
    if args.backend == "nccl" and not use_gpu:
        raise RuntimeError("launch_tiles: backend nccl needs --device cuda")
    if use_gpu:
        if not torch.cuda.is_available():
            raise RuntimeError("launch_tiles: --device cuda needs a HIP device (use --backend gloo --device cpu for CPU-only dry runs)")
        torch.cuda.set_device(local % torch.cuda.device_count())
        device = torch.device("cuda", torch.cuda.current_device())
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    ctl_dev = device if args.backend == "nccl" else None            # where the control collectives' tensors live
    entry = resolve_entry(args.entry)
    names = tiles.list_tiles(args.data)
    out_names = tile_dirnames(len(names))
    mine = tiles.assign_tiles(len(names), world, rank)
    tiles.barrier(device)
    t0 = time.perf_counter()
    iters = 0
    done = []
    for i in mine:
        paths = tiles.tile_output_paths(args.output, out_names[i])
        for p in paths.values():
            os.makedirs(p, exist_ok=True)
        iters += int(entry(os.path.join(args.data, names[i]), paths, device, i))
        done.append(i)
    tiles.barrier(device)
    elapsed = time.perf_counter() - t0
    t_max, n_sum = tiles.reduce_job(elapsed, iters, ctl_dev)
    summary = None
    if rank == 0:
        summary = {"tiles": len(names), "workers": world, "elapsed_s": t_max, "iterations": n_sum,
                   "iters_per_s": (n_sum / t_max) if t_max > 0 else 0.0, "backend": args.backend if world > 1 else None,
                   "device": device.type}
        print(json.dumps(summary), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return summary


def hip_device_bdfs(sysfs_root="/"):
    """PCI addresses ("dddd:bb:dd.f") of the GPUs in HIP's UNFILTERED device order, from the KFD topology (/sys/class/kfd/kfd/topology/nodes/<n>/
    properties: the GPU nodes -- simd_count > 0 -- in node order are what ROCr / HIP enumerate; `domain` and `location_id` = bus << 8 | devfn give
    the address).  /sys/class/drm/card* order is NOT that order in general.  Also returns each GPU's `unique_id` (HIP_VISIBLE_DEVICES accepts
    "GPU-<unique_id hex>").  -> list of (bdf, unique_id)."""
    base = os.path.join(sysfs_root, "sys/class/kfd/kfd/topology/nodes")
    out = []
    for n in sorted((d for d in os.listdir(base) if d.isdigit()), key=int):
        props = {}
        try:
            with open(os.path.join(base, n, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v.strip()
        except OSError:
            continue                                   # a GPU of the node this container may not open (device cgroup): HIP does not enumerate it either
        if int(props.get("simd_count", "0")) <= 0:
            continue                                   # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        out.append((f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}", int(props.get("unique_id", "0"))))
    return out


def gpu_numa_cpus(visible_id, sysfs_root="/"):
    """CPUs of the NUMA node of the GPU that `visible_id` names -- an entry of HIP_VISIBLE_DEVICES as a child will receive it: an index into HIP's
    unfiltered enumeration, or "GPU-<unique_id hex>" -- resolved by PCI address (/sys/bus/pci/devices/<bdf>/numa_node), or None when that
    cannot be told.  (Round 3 took the rank's ordinal and the order of /sys/class/drm/card*: with HIP_VISIBLE_DEVICES=4,5,6,7 rank 0 ran on GPU 4
    with GPU 0's cores.)"""
    try:
        gpus = hip_device_bdfs(sysfs_root)
        vid = str(visible_id).strip()
        if vid.upper().startswith("GPU-"):
            want = int(vid[4:], 16)
            match = [b for b, u in gpus if u == want]
            if not match:
                return None
            bdf = match[0]
        else:
            bdf = gpus[int(vid)][0]
        node = int(open(os.path.join(sysfs_root, "sys/bus/pci/devices", bdf, "numa_node")).read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open(os.path.join(sysfs_root, f"sys/devices/system/node/node{node}/cpulist")).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus or None
    except Exception:
        return None


def rank_placement(ranks, gpus, visible=None, sysfs_root="/", affinity=True):
    """-> [(HIP_VISIBLE_DEVICES value, cpu set or None)] per rank: rank r works on the (r % gpus)-th entry of the launcher's own HIP_VISIBLE_DEVICES
    (`visible`; all devices 0..gpus-1 when unset) and is pinned to the cores of THAT device's NUMA node."""
    ids = [v.strip() for v in visible.split(",") if v.strip() != ""] if visible else [str(i) for i in range(gpus)]
    out = []
    for r in range(ranks):
        dev = ids[(r % gpus) % len(ids)]
        out.append((dev, gpu_numa_cpus(dev, sysfs_root) if affinity else None))
    return out


def worker_device(args):
    """'cuda' | 'cpu': where the tiles train.  Follows --device when given, else the control backend (nccl -> cuda, gloo -> cpu)."""
    d = getattr(args, "device", None)
    return d if d else ("cuda" if args.backend == "nccl" else "cpu")


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(cmd_of_rank, ranks, gpus, port, pin_gpus=True, affinity=True, extra_env=None, poll_s=0.05):
    """Start `ranks` child processes with the torchrun environment contract (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*); rank r works on GPU
    r % gpus.  With `pin_gpus` every child is pinned: HIP_VISIBLE_DEVICES = its GPU (the process sees exactly one device and LOCAL_RANK is
    0, so LOCAL_RANK-independent code paths cannot land on GPU 0 by accident) and, where sysfs tells the GPU's NUMA node, CPU affinity to
    that node's cores.  The children are polled: when one exits non-zero the others are terminated instead of waiting in a collective
    until the backend times out.  Returns the first non-zero exit code, or 0.  Used by the tile launcher and by `bench.py --gpus N`."""
    procs = []
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # the directory that holds gsrast/ and the drop-in packages
    place = rank_placement(ranks, gpus, os.environ.get("HIP_VISIBLE_DEVICES"), affinity=affinity)
    for r in range(ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if pin_gpus else str(r), WORLD_SIZE=str(ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        if extra_env:
            env.update(extra_env)
        if pin_gpus:
            env["HIP_VISIBLE_DEVICES"] = place[r][0]
        cpus = place[r][1] if pin_gpus else None
        pre = (lambda c=cpus: os.sched_setaffinity(0, c)) if cpus else None
        procs.append(subprocess.Popen(cmd_of_rank(r), env=env, preexec_fn=pre))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(poll_s)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:              # a dead rank leaves the others blocked in barrier / init_process_group
                    q.terminate()
    return rc


def spawn(args):
    """Parent of the tile job: gpus x workers-per-gpu children through spawn_ranks.  When several workers share a GPU the children get
    the gloo control backend (RCCL rejects two ranks on one device) and keep --device cuda."""
    per = max(1, int(getattr(args, "workers_per_gpu", 1)))
    n = args.gpus * per                                                            # ranks; rank r works on GPU r % gpus
    device = worker_device(args)
    backend = args.backend
    if backend == "nccl" and per > 1:
        backend = "gloo"
        print(f"launch_tiles: {per} workers per GPU -> control collectives over gloo (host tensors), tiles stay on the HIP devices", file=sys.stderr)
    cmd = [sys.executable, "-m", "gsrast.launch_tiles", "--data", args.data, "--output", args.output, "--entry", args.entry,
           "--backend", backend, "--device", device, "--gpus", str(args.gpus), "--workers-per-gpu", str(per), "--_child"]
    return spawn_ranks(lambda r: cmd, n, args.gpus, args.port, pin_gpus=(device == "cuda"), affinity=not args.no_affinity)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--data", required=True, help="partitioned scene directory holding tile_* sub-directories")
    ap.add_argument("--output", required=True)
    ap.add_argument("--entry", required=True, help="module:function(tile_dir, out_paths, device, tile_index) -> iterations")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="backend of the CONTROL collectives (barrier, job reduction); nccl = RCCL, one rank per GPU")
    ap.add_argument("--device", default=None, choices=["cuda", "cpu"], help="where the tiles train (default: cuda for nccl, cpu for gloo)")
    ap.add_argument("--port", type=int, default=29531)
    ap.add_argument("--workers-per-gpu", type=int, default=1, help="concurrent tile workers per GPU (2 fills the launch gaps of one: +20 %% aggregate it/s)")
    ap.add_argument("--no-affinity", action="store_true", help="do not pin workers to the CPUs of their GPU's NUMA node")
    ap.add_argument("--_child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args._child or "RANK" in os.environ or args.gpus * max(1, args.workers_per_gpu) == 1:
        worker(args)
        return 0
    return spawn(args)


if __name__ == "__main__":
    sys.exit(main())
