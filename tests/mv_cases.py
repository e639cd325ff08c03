"""Seeded two-view inputs for the PGSR multi-view losses: two cameras looking at a textured world plane, analytic plane depth / normal /
distance / gray texture, with a smooth depth perturbation so that the reprojection error straddles the noise threshold."""
import math
import types

import numpy as np

import scenes


def _view(W, H, yaw, t, n_w, c_w, phase, amp, fscale=0.9, tex=1.0):
    cam = scenes.make_camera(W, H, fscale * W, fscale * W, yaw_deg=yaw, t=t)
    wvt = cam["viewmatrix"].astype(np.float64)
    R, T = wvt[:3, :3].copy(), wvt[3, :3].copy()
    c = dict(R=R, T=T, Fx=fscale * W, Fy=fscale * W, Cx=0.5 * W, Cy=0.5 * H)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    ray = np.stack([(xx - c["Cx"]) / c["Fx"], (yy - c["Cy"]) / c["Fy"], np.ones_like(xx)], -1)
    n_c = n_w @ R; c_c = c_w + n_c @ T
    d0 = c_c / (ray @ n_c)
    Xw = (ray * d0[..., None] - T) @ R.T
    a, b = Xw[..., 0], Xw[..., 1]
    gray = 0.5 + 0.2 * np.sin(5 * tex * a) + 0.2 * np.cos(7 * tex * b) + 0.1 * np.sin(11 * tex * (a + b))
    depth = d0 * (1 + amp * np.sin(xx / 5.0 + phase) * np.cos(yy / 4.0))
    sgn = -1.0 if (ray[H // 2, W // 2] @ n_c) > 0 else 1.0
    return c, depth.astype(np.float32), (sgn * n_c).astype(np.float32), float(abs(c_c)), gray.astype(np.float32)


def plane_pair(W=72, H=54, seed=0, amp=0.12, near_yaw=-5.0, near_t=(-0.25, 0.04, 0.03), tex=1.0):
    r = np.random.default_rng(seed)
    n_w = np.array([0.12, -0.2, -1.0]) + r.normal(0, 0.05, 3); n_w /= np.linalg.norm(n_w)
    c_w = float(n_w @ np.array([0.0, 0.0, 3.0]))
    vc, dv, nv, distv, gv = _view(W, H, 4.0, (0.05, 0.0, 0.0), n_w, c_w, 0.0, amp, tex=tex)
    nc, dn, _, _, gn = _view(W, H, near_yaw, near_t, n_w, c_w, 1.3, amp, tex=tex)
    normal = (nv[:, None, None] + r.normal(0, 0.01, (3, H, W))).astype(np.float32)
    dist = (distv * (1 + r.normal(0, 0.002, (H, W)))).astype(np.float32)
    return dict(W=W, H=H, view=vc, near=nc, plane_depth=dv[None], near_plane_depth=dn[None], rendered_normal=normal, rendered_distance=dist[None],
                gray=gv[None], near_gray=gn[None])


def cam_ns(c):
    return types.SimpleNamespace(**c, ncc_scale=1.0)
