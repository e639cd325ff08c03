"""Synthetic RGB-D frames for the TSDF tests (numpy only): a wavy surface seen from a few nearby poses."""
import numpy as np


def frames(n=4, W=160, H=120, seed=0, holes=True):
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 140.0, 135.0, W / 2 - 0.5, H / 2 - 0.5
    out = []
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    for k in range(n):
        a = 0.06 * k
        E = np.array([[np.cos(a), 0, np.sin(a), 0.05 * k], [0, 1, 0, -0.03 * k], [-np.sin(a), 0, np.cos(a), 0.08 * k], [0, 0, 0, 1]], np.float64)
        depth = (4.0 + 0.4 * np.sin(u / 17.0 + k) + 0.3 * np.cos(v / 13.0)).astype(np.float32)[None]
        if holes:
            depth[0, : 6 + k] = 0.0                       # masked rows (mesh_utils.py:165-166 zeroes depth where the alpha mask is low)
            depth[0, 40:50, 60:90] = 9.0                  # beyond depth_trunc
        rgb = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
        out.append(dict(rgb=rgb, depth=depth, fx=fx, fy=fy, cx=cx, cy=cy, E=E.astype(np.float32)))
    return out


def rgb8(rgb):
    return (np.clip(rgb, 0, 1) * 255).astype(np.uint8).astype(np.float32)
