"""Drop-in replacement for `diff_surfel_rasterization` (submodules/diff-surfel-rasterization/
diff_surfel_rasterization/__init__.py): 2DGS surfel rasterizer with the fork's 11-channel auxiliary map.

    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                                        rotations=None, cov3D_precomp=None) -> (color[3,H,W], radii[P], allmap[11,H,W])

allmap channels: 0 depth*alpha, 1 alpha, 2-4 view-space normal, 5 median depth, 6 distortion, 7 median splat index,
8-10 median normal.  `scales` is (P,2); the `cov3D_precomp` slot carries a precomputed transMat (P,9).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from gsrast import SURFEL
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
            cpu_args = _rz.cpu_deep_copy_tuple(args)
            try:
                num_rendered, outs, radii, geomBuffer, binningBuffer, imgBuffer = _rz.forward(SURFEL, *args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, outs, radii, geomBuffer, binningBuffer, imgBuffer = _rz.forward(SURFEL, *args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                              geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)     # an output the loss does not use arrives as None (a null pointer for the kernels), not as a zero-filled image
        return outs["color"], radii, outs["others"]

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
         geomBuffer, binningBuffer, imgBuffer) = ctx.saved_tensors
        pos = (SURFEL, ctx.num_rendered, rs, radii, means3D, sh, colors_precomp, opacities, scales, rotations,
               cov3Ds_precomp, None, geomBuffer, binningBuffer, imgBuffer)
        kw = dict(grad_color=grad_out_color, grad_others=grad_depth)
        if rs.debug:
            cpu_args = _rz.cpu_deep_copy_tuple(pos[3:] + (grad_out_color, grad_depth))
            try:
                g = _rz.backward(*pos, **kw)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            g = _rz.backward(*pos, **kw)
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
        from diff_gaussian_rasterization import GaussianRasterizer as _G
        return _G(self.raster_settings).markVisible(positions)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        dev = means3D.device
        if shs is None:
            shs = torch.empty(0, device=dev)
        if colors_precomp is None:
            colors_precomp = torch.empty(0, device=dev)
        if scales is None:
            scales = torch.empty(0, device=dev)
        if rotations is None:
            rotations = torch.empty(0, device=dev)
        if cov3D_precomp is None:
            cov3D_precomp = torch.empty(0, device=dev)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   raster_settings)
