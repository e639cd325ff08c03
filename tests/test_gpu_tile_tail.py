"""GPU test (pytest -m gpu) of BASELINE config 5's tail, end to end on the device: for each of two scene tiles, every camera inside the tile's box is
rendered by diff_surfel_rasterization (HIP), the surface depth is formed as gssr/scene/twodgs_scene.py:96-111 does, fed WITHOUT A HOST COPY to
gsrast.tsdf.ScalableTSDFVolume.integrate, and the two tiles' volumes are merged (merge_from) -- /root/reference/extract_mesh_split.py:91-119,
gssr/utils/mesh_utils.py:108-121,154-178.  Compared voxel for voxel with {oracle rasterizer -> oracle sparse TSDF} over all frames in one volume:
same units, weights exact, tsdf within 1e-4.  Depth pixels whose gate decisions can flip under float32 rounding (oracle.Truth margins) are masked
to 0 on both sides through the reference's own alpha-mask path (mesh_utils.py:165-166), so that the comparison is over gate-robust depth."""
import numpy as np
import pytest
import torch

import oracle
import tile_tail
import tsdf_cases

pytestmark = pytest.mark.gpu

VL, TR = 0.05, 0.25


@pytest.mark.parametrize("depth_ratio", [0.0, 1.0])
def test_two_tiles_rendered_fused_and_merged_on_the_device(depth_ratio):
    import diff_surfel_rasterization as dsr
    import hiprun
    from gsrast.tsdf import ScalableTSDFVolume
    tiles = tile_tail.make_tiles(n_tiles=2, cams_per_tile=3, P=6000, W=320, H=208, seed=3)
    DT = 25.0
    ref = oracle.SparseTSDF(VL, TR)
    vols = []
    n_masked = n_px = 0
    for tile in tiles:
        vol = ScalableTSDFVolume(VL, TR, capacity_units=4096)
        for k, cam in enumerate(tile["cams"]):
            sc = tile_tail.frame_scene(tile, k)
            fx, fy, cx, cy, E = tile_tail.o3d_camera(cam)
            # ---- oracle side: float32 CPU rasterizer, its float64 twin for the gate margins, numpy depth, CPU sparse volume
            with oracle.Forward(sc, "surfel") as f:
                with oracle.Truth(sc, "surfel", f) as t:
                    fragile = t.fragile()
                d_ref = tile_tail.surf_depth_np(f.others, depth_ratio)
                d_ref[0][fragile] = 0.0
                ref.integrate(tsdf_cases.rgb8(f.color), d_ref, fx, fy, cx, cy, E, depth_trunc=DT)
            n_masked += int(fragile.sum()); n_px += fragile.size
            # ---- HIP side: drop-in rasterizer -> depth -> volume, all device tensors
            t_ = hiprun.to_dev(sc, "cuda")
            rs = hiprun.settings("surfel", t_)
            with torch.no_grad():
                color, radii, allmap = dsr.GaussianRasterizer(rs)(means3D=t_["means3D"], means2D=torch.zeros_like(t_["means3D"]), opacities=t_["opacities"],
                                                                 colors_precomp=t_["colors_precomp"], scales=t_["scales"], rotations=t_["rotations"])
                depth = tile_tail.surf_depth_torch(allmap, depth_ratio)
                depth.masked_fill_(torch.from_numpy(fragile).to(depth.device)[None], 0.0)          # the reference's alpha-mask path (mesh_utils.py:165-166)
            assert color.is_cuda and depth.is_cuda
            vol.integrate(color, depth, fx, fy, cx, cy, E, depth_trunc=DT)
        assert vol.num_units > 20
        vols.append(vol)
    assert n_masked < 0.08 * n_px                                    # the mask removes the fragile pixels, not the image
    joint = ScalableTSDFVolume(VL, TR, capacity_units=2 * sum(v.num_units for v in vols))      # merge_from does not grow the pool
    for v in vols:
        joint.merge_from(v)
    assert joint.num_units >= max(v.num_units for v in vols)
    got = tuple(x.cpu().numpy() for x in joint.units())
    rep = tile_tail.compare_units(got, ref.units(), tsdf_tol=1e-4, color_tol=1.0, bad_frac=1e-3)
    assert rep["units"] > 100 and (got[2] > 0).sum() > 10000, rep
