"""Drop-in replacement for `diff_plane_rasterization` (submodules/diff-plane-rasterization/
diff_plane_rasterization/__init__.py): PGSR plane rasterizer.

    GaussianRasterizationSettings(..., prefiltered, render_geo, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, means2D_abs, opacities, shs=None, colors_precomp=None,
                                        scales=None, rotations=None, cov3D_precomp=None, all_map=None)
        -> (color[3,H,W], radii[P], out_observe[P] int32, out_all_map[5,H,W], plane_depth[1,H,W])
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from gsrast import PLANE
from gsrast import rasterize as _rz


def rasterize_gaussians(means3D, means2D, means2D_abs, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, all_map, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, means2D_abs, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, all_map, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, means2D_abs, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                all_maps, raster_settings):
        args = (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, all_maps, raster_settings)
        if raster_settings.debug:
            cpu_args = _rz.cpu_deep_copy_tuple(args)
            try:
                num_rendered, outs, radii, geomBuffer, binningBuffer, imgBuffer = _rz.forward(PLANE, *args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, outs, radii, geomBuffer, binningBuffer, imgBuffer = _rz.forward(PLANE, *args)
        out_observe, out_all_map, out_plane_depth = outs["observe"], outs["all_map"], outs["plane_depth"]
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(out_all_map, colors_precomp, all_maps, means3D, scales, rotations, cov3Ds_precomp, radii,
                              sh, opacities, geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii, out_observe)
        ctx.set_materialize_grads(False)     # an output the loss does not use arrives as None (a null pointer for the kernels), not as a zero-filled image
        return outs["color"], radii, out_observe, out_all_map, out_plane_depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_out_observe, grad_out_all_map, grad_out_plane_depth):
        rs = ctx.raster_settings
        (all_map_pixels, colors_precomp, all_maps, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
         geomBuffer, binningBuffer, imgBuffer) = ctx.saved_tensors
        pos = (PLANE, ctx.num_rendered, rs, radii, means3D, sh, colors_precomp, opacities, scales, rotations,
               cov3Ds_precomp, all_maps, geomBuffer, binningBuffer, imgBuffer)
        kw = dict(grad_color=grad_out_color, grad_all_map=grad_out_all_map, grad_plane_depth=grad_out_plane_depth,
                  all_map_pixels=all_map_pixels)
        if rs.debug:
            cpu_args = _rz.cpu_deep_copy_tuple(pos[3:] + (grad_out_color, grad_out_all_map, grad_out_plane_depth))
            try:
                g = _rz.backward(*pos, **kw)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            g = _rz.backward(*pos, **kw)
        # (means3D, means2D, means2D_abs, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, all_map, settings)
        return (g["dL_dmeans3D"], g["dL_dmeans2D"], g["dL_dmeans2D_abs"], g["dL_dsh"], g["dL_dcolors"],
                g["dL_dopacity"], g["dL_dscales"], g["dL_drotations"], g["dL_dcov3D"], g["dL_dall_map"], None)


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
    render_geo: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        from diff_gaussian_rasterization import GaussianRasterizer as _G
        return _G(self.raster_settings).markVisible(positions)

    def forward(self, means3D, means2D, means2D_abs, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, all_map=None):
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
        if all_map is None:
            all_map = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, means2D_abs, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, all_map, raster_settings)
