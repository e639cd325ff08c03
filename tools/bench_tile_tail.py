"""Throughput of BASELINE config 5's tail on one MI355X: per frame, diff_surfel_rasterization forward (HIP) -> surface depth
(twodgs_scene.py:96-111) -> ScalableTSDFVolume.integrate, every image a device tensor from the rasterizer to the volume; then the merge of the
tiles' volumes (extract_mesh_split.py:91-119, mesh_utils.py:108-121,154-178).  The reference copies rgb and depth of every frame to the host, hands
them to Open3D on the CPU and builds one volume.

    python tools/bench_tile_tail.py [--tiles 2 --cams 8 --P 300000 --W 1920 --H 1080] > profiles/r04_tile_tail.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gs-sr_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import hiprun  # noqa: E402
import tile_tail  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=2); ap.add_argument("--cams", type=int, default=8)
    ap.add_argument("--P", type=int, default=300000); ap.add_argument("--W", type=int, default=1920); ap.add_argument("--H", type=int, default=1080)
    ap.add_argument("--mesh-res", type=int, default=1024); ap.add_argument("--depth-trunc", type=float, default=8.0)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--capacity", type=int, default=65536, help="initial units per tile volume (the cloud scene needs ~88k: the pool grows on the way)")
    ap.add_argument("--count-updates", action="store_true", help="one extra (untimed) pass that counts the voxels every frame updates (from the weights)")
    ap.add_argument("--surface", default="cloud", choices=["cloud", "terrain"], help="cloud: SURVEY 8d's independent surfels (depth maps of noise: every line of every listed "
                    "unit is touched); terrain: the same surfels on one smooth surface (what a trained scene renders)")
    ap.add_argument("--sync-frames", action="store_true", help="integrate(defer=False): one host synchronisation per frame, as in rounds 2-4")
    a = ap.parse_args()
    import diff_surfel_rasterization as dsr
    from gsrast.tsdf import ScalableTSDFVolume
    tiles = tile_tail.make_tiles(a.tiles, a.cams, a.P, a.W, a.H, seed=0, sigma_px=4.0, surface=a.surface)
    vl = a.depth_trunc / a.mesh_res                     # extract_mesh_split.py:82-84: voxel = depth_trunc / mesh_res, sdf_trunc = 5 voxels
    tr = 5.0 * vl
    frames = []
    for tile in tiles:
        fr = []
        for k, cam in enumerate(tile["cams"]):
            t = hiprun.to_dev(tile_tail.frame_scene(tile, k), "cuda")
            fr.append((hiprun.settings("surfel", t), t, tile_tail.o3d_camera(cam)))
        frames.append(fr)

    touched = []

    updated = []

    def run(count=False):
        vols, n_units, evs = [], [], []
        t_r = t_i = 0.0
        touched.clear(); updated.clear()
        for fr in frames:
            vol = ScalableTSDFVolume(vl, tr, capacity_units=a.capacity)
            for rs, t, (fx, fy, cx, cy, E) in fr:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                with torch.no_grad():
                    color, radii, allmap = dsr.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"],
                                                                     colors_precomp=t["colors_precomp"], scales=t["scales"], rotations=t["rotations"])
                    depth = tile_tail.surf_depth_torch(allmap, 0.0)
                e1.record()
                if count:
                    w0 = vol.weight_sum()
                vol.integrate(color, depth, fx, fy, cx, cy, E, depth_trunc=a.depth_trunc, defer=not a.sync_frames and not count)
                e2.record()
                evs.append((e0, e1, e2, vol))
                if count:
                    updated.append(vol.weight_sum() - w0); touched.append(vol.last_touched)
            vol.finish()
            vols.append(vol); n_units.append(vol.num_units)
        torch.cuda.synchronize()
        for e0, e1, e2, _ in evs:
            t_r += e0.elapsed_time(e1); t_i += e1.elapsed_time(e2)
        t0 = time.perf_counter()
        joint = ScalableTSDFVolume(vl, tr, capacity_units=max(a.capacity, 2 * sum(n_units)))
        for v in vols:
            joint.merge_from(v)
        torch.cuda.synchronize()
        return t_r, t_i, (time.perf_counter() - t0) * 1e3, n_units, joint.num_units

    run()                                               # warm-up: arenas, capacity hints, pool growth
    best = None
    for _ in range(a.repeat):
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        r = run()
        wall = (time.perf_counter() - w0) * 1e3
        if best is None or wall < best[0]:
            best = (wall,) + r
    wall, t_r, t_i, t_m, n_units, n_joint = best
    nf = a.tiles * a.cams
    upd = units = None
    if a.count_updates:
        run(count=True)
        upd = sum(updated) / max(len(updated), 1); units = sum(touched) / max(len(touched), 1)
    print(json.dumps({"what": "config 5 tail: surfel render -> surface depth -> sparse TSDF integrate per frame, then merge of the tiles' volumes; device-resident images",
                      "device": torch.cuda.get_device_name(0), "surface": a.surface, "tiles": a.tiles, "cameras_per_tile": a.cams, "gaussians_per_tile": a.P, "image": [a.W, a.H],
                      "voxel_length": vl, "sdf_trunc": tr, "depth_trunc": a.depth_trunc, "frames": nf,
                      "render_ms_per_frame": round(t_r / nf, 4), "integrate_ms_per_frame": round(t_i / nf, 4),
                      "units_integrated_per_frame": None if units is None else round(units), "voxels_updated_per_frame": None if upd is None else round(upd),
                      "updated_fraction_of_the_listed_units": None if upd is None else round(upd / (units * 4096), 3),
                      "algorithmic_GB_per_frame": None if upd is None else round((upd * 40 + a.W * a.H * 16) / 1e9, 3),
                      "frames_deferred": not a.sync_frames,
                      "integrate_note": "HIP events around ScalableTSDFVolume.integrate: texels, touch, stamp, voxel pass and any pool growth (--capacity units per tile at the start; "
                                        "growth adds a chunk of records, no voxel is copied); deferred frames are only enqueued (up to 8 in flight), --sync-frames restores the per-frame host wait",
                      "frames_per_s_gpu": round(nf / ((t_r + t_i) * 1e-3), 1), "merge_ms": round(t_m, 3), "units_per_tile": n_units, "units_merged": n_joint,
                      "wall_ms_total": round(wall, 2), "frames_per_s_wall_incl_merge": round(nf / (wall * 1e-3), 1)}))


if __name__ == "__main__":
    main()
