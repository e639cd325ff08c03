"""Block-sparse TSDF integration at the reference's DEFAULT resolution (extract_mesh.py:125-128: voxel = depth_trunc / 1024, sdf_trunc = 5 voxels):
frames per second, opened / updated units per frame, the voxels a frame actually updates (from the weights) and the algorithmic bandwidth on THOSE
(40 B per updated voxel + 16 B per pixel) from HIP events around every integrate call.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd"))
from gsrast.tsdf import ScalableTSDFVolume      # noqa: E402


def main():
    W, H = 1920, 1080
    depth_trunc = 8.0
    vl = depth_trunc / 1024
    vol = ScalableTSDFVolume(vl, 5 * vl, capacity_units=1 << 17)
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    frames = []
    for k in range(12):
        depth = (4.0 + 0.4 * np.sin(u / 160.0 + 0.3 * k) + 0.3 * np.cos(v / 120.0)).astype(np.float32)[None]
        rgb = np.random.default_rng(k).uniform(0, 1, (3, H, W)).astype(np.float32)
        E = np.eye(4, dtype=np.float32); E[0, 3] = 0.03 * k
        frames.append((torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), E))
    f = 0.8 * W
    touched, ms, upd = [], [], []
    for i, (rgb, depth, E) in enumerate(frames):
        w0 = vol.weight_sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
        vol.integrate(rgb, depth, f, f, W / 2, H / 2, E, depth_trunc=depth_trunc)
        e1.record(); torch.cuda.synchronize()
        if i >= 2:
            touched.append(vol.last_touched); ms.append((e0.elapsed_time(e1), 1e3 * (time.perf_counter() - t0)))
            upd.append(vol.weight_sum() - w0)      # every updated voxel gained exactly one unit of weight
    units = float(np.mean(touched)); gpu = float(np.mean([a for a, _ in ms])); wall = float(np.mean([b for _, b in ms]))
    # the same frames once more, only ENQUEUED (integrate(defer=True): no host wait between the frames, the status of a frame is read when the next one starts)
    vol2 = ScalableTSDFVolume(vl, 5 * vl, capacity_units=1 << 17)
    evs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, (rgb, depth, E) in enumerate(frames):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vol2.integrate(rgb, depth, f, f, W / 2, H / 2, E, depth_trunc=depth_trunc, defer=True)
        e1.record(); evs.append((e0, e1))
    vol2.finish(); torch.cuda.synchronize()
    wall_d = 1e3 * (time.perf_counter() - t0) / len(frames)
    gpu_d = float(np.mean([a.elapsed_time(b) for a, b in evs[2:]]))
    assert vol2.num_units == vol.num_units
    print(json.dumps({"what": "ScalableTSDFVolume.integrate, 1920x1080 depth+colour frames, voxel = depth_trunc/1024 = 7.8 mm, sdf_trunc = 5 voxels, stride 4",
                      "units_allocated": vol.num_units, "pool_capacity_units": vol.cap, "units_updated_per_frame": round(units),
                      "ms_per_frame_gpu_events": round(gpu, 3), "ms_per_frame_wall": round(wall, 3), "frames_per_s": round(1e3 / wall, 1),
                      "deferred_ms_per_frame_gpu_events": round(gpu_d, 3), "deferred_ms_per_frame_wall": round(wall_d, 3), "deferred_frames_per_s": round(1e3 / wall_d, 1),
                      "voxels_updated_per_frame": round(float(np.mean(upd))), "updated_fraction_of_the_listed_units": round(float(np.mean(upd)) / (units * 4096), 3),
                      "algorithmic_bytes_per_frame": int(np.mean(upd) * 40 + W * H * 16),
                      "algorithmic_GBps": round((np.mean(upd) * 40 + W * H * 16) / (gpu * 1e-3) / 1e9, 1),
                      "algorithmic_note": "SURVEY 8d: 40 B per UPDATED voxel (tsdf, weight, 3 colour floats read + written) + 16 B per pixel; round 4 counted every voxel of a listed unit",
                      "note": "the three launches of one frame (touch-insert, stamp, integrate) + the 16-byte counter read-back are inside the event pair; "
                              "a dense grid of this resolution would be >= 1024^3 voxels = 21 GB"}))


if __name__ == "__main__":
    main()
