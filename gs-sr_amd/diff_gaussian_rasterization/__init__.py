"""Drop-in replacement for the reference's CUDA extension `diff_gaussian_rasterization`
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py), backed by the hand-written HIP
library libgsrast_hip.so for MI355X (gfx950).  Same names, argument meaning, return tuples and error messages:

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
                                  projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                                        rotations=None, cov3D_precomp=None) -> (color[3,H,W], radii[P] int32)
    GaussianRasterizer.markVisible(positions) -> bool[P]
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from gsrast import EWA, lib, check, ptr, stream_ptr, dev_f32
from gsrast import rasterize as _rz


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        args = (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, None, raster_settings)
        if raster_settings.debug:
            cpu_args = _rz.cpu_deep_copy_tuple(args)   # copy them before they can be corrupted
            try:
                num_rendered, outs, radii, geomBuffer, binningBuffer, imgBuffer = _rz.forward(EWA, *args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, outs, radii, geomBuffer, binningBuffer, imgBuffer = _rz.forward(EWA, *args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                              geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)     # an output the loss does not use arrives as None (a null pointer for the kernels), not as a zero-filled image
        return outs["color"], radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
         geomBuffer, binningBuffer, imgBuffer) = ctx.saved_tensors
        kw = dict(grad_color=grad_out_color)
        pos = (EWA, ctx.num_rendered, rs, radii, means3D, sh, colors_precomp, opacities, scales, rotations,
               cov3Ds_precomp, None, geomBuffer, binningBuffer, imgBuffer)
        if rs.debug:
            cpu_args = _rz.cpu_deep_copy_tuple(pos[3:] + (grad_out_color,))
            try:
                g = _rz.backward(*pos, **kw)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            g = _rz.backward(*pos, **kw)
        # (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)
        return (g["dL_dmeans3D"], g["dL_dmeans2D"], g["dL_dsh"], g["dL_dcolors"], g["dL_dopacity"], g["dL_dscales"],
                g["dL_drotations"], g["dL_dcov3D"], None)


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

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            rs = self.raster_settings
            pos = dev_f32(positions, "positions", allow_empty=True)
            P = int(positions.size(0))
            visible = torch.zeros((P,), dtype=torch.bool, device=positions.device)
            if P:
                vm, pm = dev_f32(rs.viewmatrix, "viewmatrix"), dev_f32(rs.projmatrix, "projmatrix")
                check(lib().gsr_mark_visible(P, ptr(pos), ptr(vm), ptr(pm), ptr(visible), stream_ptr(positions.device)),
                      "mark_visible")
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   raster_settings)
