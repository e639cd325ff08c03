"""Point-cloud interchange with the reference: the `point_cloud.ply` files its Gaussian models write and read.

The reference goes through the `plyfile` package (`PlyData([PlyElement.describe(elements, 'vertex')]).write(path)`): one `vertex` element whose
properties are all `float` (numpy 'f4'), native byte order, binary.  This module writes and parses that container directly with numpy
-- no third-party dependency -- and provides the two property layouts of the reference's models:
  * explicit Gaussians  (`VanillaGaussian.save_gaussians / load_gaussians`, gssr/gaussian/vanilla_gaussian.py:140-214),
  * anchors             (`ScaffoldGaussian.save_gaussians / load_gaussians`, gssr/gaussian/scaffold_gaussian.py:388-456, and
                         `OctreeGaussian`'s, octree_gaussian.py:276-360, with its `level` / `extra_level` / `info` columns).
Tensors in, tensors out (float32, on the device asked for); the (N, C, S) <-> flat column conventions -- `transpose(1, 2).flatten(1)` on
write, `reshape(N, 3, -1)` + `transpose(1, 2)` on read -- are the reference's.
"""
import numpy as np
import torch

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def write_vertex_table(path, names, table):
    """One `vertex` element, every property `float`: the file `plyfile` produces for a structured array of 'f4' fields.
    names: property names in column order; table: (N, len(names)) array-like."""
    a = np.ascontiguousarray(np.asarray(table, dtype="<f4"))
    if a.ndim != 2 or a.shape[1] != len(names):
        raise ValueError(f"gsrast.ply: table shape {a.shape} does not match {len(names)} property names")
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {a.shape[0]}"] + [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        f.write(a.tobytes())


def read_vertex_table(path):
    """-> (names, float64 (N, len(names)) array) of the first element of a PLY file (binary little/big endian or ascii; scalar properties only)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"gsrast.ply: {path} is not a PLY file")
        fmt, count, props, in_first, seen = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError("gsrast.ply: end of file inside the header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen += 1
                in_first = seen == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError("gsrast.ply: list properties are not part of the reference's point-cloud files")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError("gsrast.ply: header without format / element")
        names = [n for n, _ in props]
        if fmt == "ascii":
            rows = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2) if count else np.zeros((0, len(names)))
            return names, rows.reshape(count, len(names))
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        rec = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
        return names, np.stack([rec[n].astype(np.float64) for n in names], axis=1) if names else np.zeros((count, 0))


def _numbered(names, prefix):
    """Columns whose name starts with `prefix`, ordered by their trailing integer (the reference sorts them the same way)."""
    sel = [(int(n.split("_")[-1]), i) for i, n in enumerate(names) if n.startswith(prefix)]
    return [i for _, i in sorted(sel)]


def _np(t):
    return t.detach().to("cpu", torch.float32).numpy()


def gaussian_attributes(n_dc, n_rest, n_scale, n_rot):
    """`VanillaGaussian.construct_list_of_attributes` (vanilla_gaussian.py:140-152)."""
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] + ["opacity"]
            + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)])


def save_gaussians(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """Explicit Gaussians, raw (pre-activation) parameters as the model holds them: xyz (N,3), features_dc (N,1,3), features_rest (N,S-1,3),
    opacity (N,1), scaling (N,2|3), rotation (N,4)."""
    x = _np(xyz)
    dc = _np(features_dc.transpose(1, 2).flatten(start_dim=1))
    rest = _np(features_rest.transpose(1, 2).flatten(start_dim=1))
    cols = np.concatenate((x, np.zeros_like(x), dc, rest, _np(opacity).reshape(-1, 1), _np(scaling), _np(rotation)), axis=1)
    write_vertex_table(path, gaussian_attributes(dc.shape[1], rest.shape[1], scaling.shape[1], rotation.shape[1]), cols)


def load_gaussians(path, max_sh_degree, device="cpu"):
    """-> dict(xyz, features_dc (N,1,3), features_rest (N,(D+1)^2-1,3), opacity (N,1), scaling, rotation) float32 on `device`."""
    names, a = read_vertex_table(path)
    col = {n: i for i, n in enumerate(names)}
    rest_idx = _numbered(names, "f_rest_")
    if len(rest_idx) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"gsrast.ply: {len(rest_idx)} f_rest columns do not belong to sh degree {max_sh_degree}")
    t = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float32, device=device)
    n = a.shape[0]
    dc = a[:, [col["f_dc_0"], col["f_dc_1"], col["f_dc_2"]]].reshape(n, 3, 1)
    rest = a[:, rest_idx].reshape(n, 3, (max_sh_degree + 1) ** 2 - 1)
    return {"xyz": t(a[:, [col["x"], col["y"], col["z"]]]), "features_dc": t(dc).transpose(1, 2).contiguous(),
            "features_rest": t(rest).transpose(1, 2).contiguous(), "opacity": t(a[:, [col["opacity"]]]),
            "scaling": t(a[:, _numbered(names, "scale_")]), "rotation": t(a[:, _numbered(names, "rot")])}


def anchor_attributes(n_offset, n_feat, n_scale, n_rot, octree=False):
    """`ScaffoldGaussian.construct_list_of_attributes` (scaffold_gaussian.py:388-399); octree=True: `OctreeGaussian`'s
    (octree_gaussian.py:276-287), which puts `level`, `extra_level`, `info` behind the normals."""
    return (["x", "y", "z", "nx", "ny", "nz"] + (["level", "extra_level", "info"] if octree else []) + [f"f_offset_{i}" for i in range(n_offset)]
            + [f"f_anchor_feat_{i}" for i in range(n_feat)] + ["opacity"] + [f"scale_{i}" for i in range(n_scale)]
            + [f"rot_{i}" for i in range(n_rot)])


def save_anchors(path, anchor, offset, anchor_feat, opacity, scaling, rotation, level=None, extra_level=None, voxel_size=None, standard_dist=None):
    """Anchors of the Scaffold / Octree models: anchor (N,3), offset (N,k,3), anchor_feat (N,32), opacity (N,1), scaling (N,6), rotation (N,4).
    Octree (`level` given): level (N,1), extra_level (N,), and the `info` column whose first two rows carry voxel_size and standard_dist
    (octree_gaussian.py:289-309)."""
    x = _np(anchor)
    off = _np(offset.transpose(1, 2).flatten(start_dim=1))
    extra_cols = []
    if level is not None:
        if voxel_size is None or standard_dist is None or len(x) < 2:
            raise ValueError("gsrast.ply: the Octree layout stores voxel_size / standard_dist in info[0] / info[1]: both are needed, and >= 2 anchors")
        info = np.zeros((len(x), 1), np.float32)
        info[0, 0] = float(voxel_size); info[1, 0] = float(standard_dist)
        el = extra_level if extra_level is not None else torch.zeros(len(x))
        extra_cols = [_np(level.float()).reshape(-1, 1), _np(el.float()).reshape(-1, 1), info]
    cols = np.concatenate([x, np.zeros_like(x)] + extra_cols + [off, _np(anchor_feat), _np(opacity).reshape(-1, 1), _np(scaling), _np(rotation)], axis=1)
    write_vertex_table(path, anchor_attributes(off.shape[1], anchor_feat.shape[1], scaling.shape[1], rotation.shape[1], level is not None), cols)


def load_anchors(path, device="cpu"):
    """-> dict(anchor, offset (N,k,3), anchor_feat, opacity (N,1), scaling, rotation) float32 on `device`; for an Octree file also
    level (N,1) int32, extra_level (N,), voxel_size, standard_dist (floats) -- octree_gaussian.py:317-327."""
    names, a = read_vertex_table(path)
    col = {n: i for i, n in enumerate(names)}
    t = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float32, device=device)
    n = a.shape[0]
    off = a[:, _numbered(names, "f_offset")].reshape(n, 3, -1)
    out = {"anchor": t(a[:, [col["x"], col["y"], col["z"]]]), "offset": t(off).transpose(1, 2).contiguous(),
           "anchor_feat": t(a[:, _numbered(names, "f_anchor_feat")]), "opacity": t(a[:, [col["opacity"]]]),
           "scaling": t(a[:, _numbered(names, "scale_")]), "rotation": t(a[:, _numbered(names, "rot")])}
    if "level" in col:
        out["level"] = torch.tensor(a[:, [col["level"]]].astype(np.int64), dtype=torch.int32, device=device)
        out["extra_level"] = t(a[:, col["extra_level"]])
        out["voxel_size"] = float(a[0, col["info"]]); out["standard_dist"] = float(a[1, col["info"]])
    return out
