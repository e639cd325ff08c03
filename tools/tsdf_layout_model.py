"""Which storage order of a 16^3 TSDF unit moves the fewest lines?  numpy model of one integrate() frame (smooth surface / blocky noise depth, three view directions):
voxels-equivalent touched per updated voxel at 16-byte group, 64-byte and 128-byte granularity for x-major columns (ABI 7) and for 4x4x4 bricks of 2x2x4 sectors (ABI 8).
The model behind EXPERIMENTS.md (77); runs on the CPU in ~2 minutes:  python tools/tsdf_layout_model.py"""
import numpy as np, sys
def run(kind, W=960, H=540, seed=0, yaw=0.0):
    rng = np.random.default_rng(seed)
    fx = W * (1600/1920); cx=(W-1)/2; cy=(H-1)/2
    vl = 8.0/1024; tr = 5*vl; UL = 16*vl
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    if kind == "smooth":
        depth = (4.0 + 0.4*np.sin(u/80.0) + 0.3*np.cos(v/60.0)).astype(np.float32)
    else:
        g = rng.uniform(1.0, 9.0, (H//5+2, W//5+2)).astype(np.float32)
        depth = g[(v//5), (u//5)]
    depth = np.where(depth > 8.0, 0, depth)
    a = np.radians(yaw); R = np.array([[np.cos(a),0,np.sin(a)],[0,1,0],[-np.sin(a),0,np.cos(a)]])  # cam->world rot
    # touch
    us, vs = u[::4, ::4].ravel(), v[::4, ::4].ravel(); d = depth[::4, ::4].ravel(); ok = d > 0
    us, vs, d = us[ok], vs[ok], d[ok]
    pc = np.stack([(us-cx)*d/fx, (vs-cy)*d/fx, d], 1); pw = pc @ R.T
    lo = np.floor((pw - tr)/UL).astype(np.int64); hi = np.floor((pw + tr)/UL).astype(np.int64)
    keys = set()
    allk = []
    for dx in (0,1):
        for dy in (0,1):
            for dz in (0,1):
                c = np.stack([np.where(dx, hi[:,0], lo[:,0]), np.where(dy, hi[:,1], lo[:,1]), np.where(dz, hi[:,2], lo[:,2])], 1)
                allk.append(c)
    units = np.unique(np.concatenate(allk), axis=0)
    n = len(units)
    # voxel pass
    ii = np.arange(16)
    X, Y, Z = np.meshgrid(ii, ii, ii, indexing="ij")   # [x,y,z]
    tot = dict(vox=0, g16=0, col64=0, col128=0, b64=0, b128=0, b256=0, s144=0)
    CH = 2000
    for s in range(0, n, CH):
        U = units[s:s+CH]
        P = (U[:, None, None, None, :]*16 + np.stack([X, Y, Z], -1)[None] + 0.5) * vl     # [m,16,16,16,3]
        pcm = P @ R     # world->cam = R^T ; p_cam = R^T p  => row-vector p @ R
        zc = pcm[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            uf = pcm[..., 0]*fx/zc + cx + 0.5; vf = pcm[..., 1]*fx/zc + cy + 0.5
        inb = (zc > 0) & (uf >= 0) & (uf < W) & (vf >= 0) & (vf < H)
        ui = np.clip(np.nan_to_num(uf), 0, W-1).astype(np.int64); vi = np.clip(np.nan_to_num(vf), 0, H-1).astype(np.int64)
        dd = depth[vi, ui]
        rx = (ui-cx)/fx; ry = (vi-cy)/fx
        sdf = (dd - zc)*np.sqrt(rx*rx+ry*ry+1)
        upd = inb & (dd > 0) & (sdf > -tr)       # [m,16,16,16]
        m = len(U)
        tot["vox"] += upd.sum()
        g = upd.reshape(m,16,16,4,4).any(-1)      # 16B groups [x,y,zq]
        tot["g16"] += g.sum()
        tot["col64"] += upd.any(-1).sum()         # z column = 64B
        tot["col128"] += upd.reshape(m,16,8,2,16).any((-1,-2)).sum()
        # bricklets: sector = 2x*2y*4z ; line = 2x*4y*4z ; brick = 4x4x4
        tot["b64"] += upd.reshape(m,8,2,8,2,4,4).any((2,4,6)).sum()
        tot["b128"] += upd.reshape(m,8,2,4,4,4,4).any((2,4,6)).sum()
        tot["b256"] += upd.reshape(m,4,4,4,4,4,4).any((2,4,6)).sum()
        tot["s144"] += upd.reshape(m,16,4,4,4,4).any((3,5)).sum()   # 1x*4y*4z slab = 64B
    V = tot["vox"]
    print(kind, "yaw", yaw, "units", n, "updated frac %.3f" % (V/(n*4096)))
    for k, b in (("g16",4),("col64",16),("col128",32),("s144",16),("b64",16),("b128",32),("b256",64)):
        print("  %-7s touched voxels-equivalent / updated = %.2f   (touched frac %.3f)" % (k, tot[k]*b/V, tot[k]*b/(n*4096)))
for kind in ("smooth", "noise"):
    for yaw in (0.0, 30.0, 90.0):
        run(kind, yaw=yaw)
