"""Generates tests/golden/*.npz from the reference's own importable Python helpers.  Run ONLY in the authoring
container (needs /root/reference); the committed .npz files are data (inputs + expected outputs), never source.

    python tests/golden/make_golden.py

Pins:
  camera_*.npz : world_view_transform / projection / full_proj / camera_center conventions
                 (gssr/utils/graphics_utils.py:38-71 getWorld2View2/getProjectionMatrix/fov2focal/focal2fov,
                  composition as gssr/cameras/__init__.py:85-88)
  sh_eval.npz  : gssr/utils/sh_utils.py:57-112 eval_sh (deg 0..3) on random coefficients/directions -- pins the SH
                 basis constants and sign conventions that 3DGS forward.cu:20-71 uses
  rgb2sh.npz   : gssr/utils/sh_utils.py RGB2SH / SH2RGB
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from gssr.utils.graphics_utils import getWorld2View2, getProjectionMatrix, fov2focal, focal2fov  # noqa: E402
from gssr.utils.sh_utils import eval_sh, RGB2SH, SH2RGB  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def cameras():
    rng = np.random.default_rng(7)
    for i, (W, H, fx) in enumerate([(1920, 1080, 1600.0), (128, 96, 106.0), (1600, 900, 1333.0)]):
        a = rng.uniform(-0.6, 0.6)
        b = rng.uniform(-0.3, 0.3)
        Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
        R = Ry @ Rx
        T = rng.uniform(-1, 1, 3)
        fovx = focal2fov(fx, W)
        fovy = focal2fov(fx, H)
        wvt = torch.tensor(getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        np.savez(os.path.join(OUT, f"camera_{i}.npz"), W=W, H=H, fx=fx, R=R, T=T, fovx=fovx, fovy=fovy,
                 focal_back=fov2focal(fovx, W), world_view_transform=wvt.numpy(), projection_matrix=proj.numpy(),
                 full_proj_transform=full.numpy(), camera_center=center.numpy())


def sh():
    g = torch.Generator().manual_seed(11)
    P = 64
    sh = torch.randn(P, 3, 16, generator=g)           # eval_sh layout: [..., C, (deg+1)^2]
    dirs = torch.randn(P, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {}
    for deg in range(4):
        out[f"deg{deg}"] = eval_sh(deg, sh[..., : (deg + 1) ** 2], dirs).numpy()
    np.savez(os.path.join(OUT, "sh_eval.npz"), sh=sh.numpy(), dirs=dirs.numpy(), **out)
    rgb = torch.rand(32, 3, generator=g)
    np.savez(os.path.join(OUT, "rgb2sh.npz"), rgb=rgb.numpy(), sh=RGB2SH(rgb).numpy(), back=SH2RGB(RGB2SH(rgb)).numpy())


if __name__ == "__main__":
    cameras()
    sh()
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
