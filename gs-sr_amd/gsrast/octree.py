"""Octree-GS level-of-detail mask fused with the visibility prefilter (include/gsrast.h gsr_octree_visible).

Replaces `OctreeGaussianModel.set_anchor_mask` (gssr/gaussian/octree_gaussian.py:255-267, incl. map_to_int_level :184-203) followed by
`OctreeScene.prefilter_voxel` (gssr/scene/octree_scene.py:136-172): no boolean-index gathers, no host synchronisation."""
import ctypes as C

import torch

from . import EWA, LodCfg, check, dev_f32, lib, make_cfg, ptr, stream_ptr

MODES = {"floor": 0, "round": 1, "ceil": 2, "progressive": 3}


def octree_visible(raster_settings, anchor, level, scaling, rotation, voxel_size, fork, standard_dist, coarse_index, dist2level="round",
                   extra_level=None, resolution_scale=1.0):
    """-> dict(anchor_mask bool[Na], visible_mask bool[Na] (== prefilter_voxel's result), radii int32[Na],
               prog_ratio float[Na,1] | None, transition_mask bool[Na] | None)   (the last two only for dist2level='progressive').
    raster_settings: the GaussianRasterizationSettings prefilter_voxel builds; scaling = get_scaling (Na,6) or (Na,3); level int (Na,) | (Na,1);
    coarse_index = `coarse_index` of set_anchor_mask (levels in use: the reference passes coarse_index - 1 to map_to_int_level)."""
    if dist2level not in MODES:
        raise ValueError(f"Unknown dist2level: {dist2level}")
    with torch.no_grad():
        a = dev_f32(anchor, "anchor", allow_empty=False)
        Na = a.shape[0]
        dev = a.device
        lv = level.reshape(-1).to(torch.int32).contiguous()
        sc = dev_f32(scaling, "scaling", allow_empty=False)
        ro = dev_f32(rotation, "rotation", allow_empty=False)
        ex = None if extra_level is None else dev_f32(extra_level.reshape(-1), "extra_level")
        amask = torch.empty(Na, dtype=torch.uint8, device=dev)
        radii = torch.empty(Na, dtype=torch.int32, device=dev)
        prog = torch.empty(Na, 1, dtype=torch.float32, device=dev) if dist2level == "progressive" else None
        trans = torch.empty(Na, dtype=torch.uint8, device=dev) if dist2level == "progressive" else None
        if Na:
            keep = []
            cfg = make_cfg(EWA, Na, raster_settings, 0, 0, False, keep)
            lod = LodCfg(float(voxel_size), float(fork), float(standard_dist), float(resolution_scale), int(coarse_index), MODES[dist2level])
            check(lib().gsr_octree_visible(C.byref(cfg), C.byref(lod), ptr(a), ptr(lv), ptr(ex), ptr(sc), int(sc.shape[1]), ptr(ro), ptr(amask),
                                           ptr(radii), ptr(prog), ptr(trans), stream_ptr(dev)), "octree_visible")
        return {"anchor_mask": amask.view(torch.bool), "visible_mask": radii > 0, "radii": radii, "prog_ratio": prog,
                "transition_mask": None if trans is None else trans.view(torch.bool)}
