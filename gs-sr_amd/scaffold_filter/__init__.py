"""Drop-in replacement for `scaffold_filter` (submodules/scaffold-filter/scaffold_filter/__init__.py): the
visibility prefilter Scaffold-GS / Octree-GS scenes run on their anchors every iteration
(gssr/scene/scaffold_scene.py:122-155, octree_scene.py:136-172).

    GaussianRasterizer(raster_settings).visible_filter(means3D, scales=None, rotations=None, cov3D_precomp=None)
        -> radii int32 [P]     (no grad)
"""
import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from gsrast import EWA, lib, check, ptr, stream_ptr, dev_f32, make_cfg


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        raster_settings = self.raster_settings
        with torch.no_grad():
            if means3D.ndimension() != 2 or means3D.size(1) != 3:
                raise RuntimeError("means3D must have dimensions (num_points, 3)")
            P = int(means3D.size(0))
            radii = torch.zeros((P,), dtype=torch.int32, device=means3D.device)
            if P == 0:
                return radii
            keep = []
            m3 = dev_f32(means3D, "means3D", allow_empty=False)
            sc = dev_f32(scales, "scales")
            if sc is not None and sc.size(1) != 3:
                sc = sc[:, :3].contiguous()
            ro = dev_f32(rotations, "rotations")
            cp = dev_f32(cov3D_precomp, "cov3D_precomp")
            if (sc is None or ro is None) and cp is None:
                raise RuntimeError("visible_filter needs scales+rotations or cov3D_precomp")
            cfg = make_cfg(EWA, P, raster_settings, 0, 0, False, keep)
            check(lib().gsr_visible_filter(C.byref(cfg), ptr(m3), ptr(sc), ptr(ro), ptr(cp), ptr(radii),
                                           stream_ptr(means3D.device)), "visible_filter")
        return radii
