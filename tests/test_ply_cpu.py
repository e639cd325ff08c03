"""Point-cloud interchange (gsrast.ply): the container plyfile writes for the reference's models, and the column conventions
of VanillaGaussian / ScaffoldGaussian / OctreeGaussian save_gaussians / load_gaussians."""
import os
import struct
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-sr_amd"))
from gsrast import ply   # noqa: E402


def test_container_is_what_plyfile_writes_for_f4_fields():
    """Header text and payload byte for byte: 'ply / format binary_little_endian 1.0 / element vertex N / property float <name> ... /
    end_header' followed by N packed little-endian float32 rows."""
    a = np.arange(12, dtype=np.float32).reshape(4, 3) * 0.5
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.ply")
        ply.write_vertex_table(p, ["x", "y", "opacity"], a)
        raw = open(p, "rb").read()
    head = b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float opacity\nend_header\n"
    assert raw[:len(head)] == head
    assert raw[len(head):] == struct.pack("<12f", *a.ravel())


def test_reader_accepts_ascii_big_endian_comments_and_mixed_types():
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "a.ply")
        open(p, "w").write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty uchar red\nproperty double z\n"
                           "element face 0\nproperty list uchar int vertex_indices\nend_header\n1.5 200 -2\n0.25 7 1e3\n")
        names, a = ply.read_vertex_table(p)
        assert names == ["x", "red", "z"] and np.array_equal(a, [[1.5, 200, -2], [0.25, 7, 1e3]])
        q = os.path.join(d, "b.ply")
        with open(q, "wb") as f:
            f.write(b"ply\nformat binary_big_endian 1.0\nelement vertex 2\nproperty float x\nproperty short k\nend_header\n")
            f.write(struct.pack(">fhfh", 1.5, -3, 2.5, 9))
        names, a = ply.read_vertex_table(q)
        assert names == ["x", "k"] and np.array_equal(a, [[1.5, -3], [2.5, 9]])
        open(p, "w").write("not a ply\n")
        with pytest.raises(ValueError):
            ply.read_vertex_table(p)


@pytest.mark.parametrize("deg,nscale", [(3, 3), (0, 2), (1, 3)])
def test_explicit_gaussians_round_trip_and_column_order(deg, nscale):
    g = torch.Generator().manual_seed(deg)
    N, S = 7, (deg + 1) ** 2
    xyz = torch.randn(N, 3, generator=g); dc = torch.randn(N, 1, 3, generator=g); rest = torch.randn(N, S - 1, 3, generator=g)
    op = torch.randn(N, 1, generator=g); sc = torch.randn(N, nscale, generator=g); rot = torch.randn(N, 4, generator=g)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "point_cloud.ply")
        ply.save_gaussians(p, xyz, dc, rest, op, sc, rot)
        names, a = ply.read_vertex_table(p)
        out = ply.load_gaussians(p, deg)
        if deg > 0:
            with pytest.raises(ValueError):
                ply.load_gaussians(p, deg - 1)
    # construct_list_of_attributes order, normals zero, f_rest channel-major: column c * (S-1) + s = features_rest[:, s, c]
    assert names == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(3 * (S - 1))] + ["opacity"]
                     + [f"scale_{i}" for i in range(nscale)] + ["rot_0", "rot_1", "rot_2", "rot_3"])
    assert np.all(a[:, 3:6] == 0)
    if S > 1:
        assert np.array_equal(a[:, 9 + 1 * (S - 1) + 0].astype(np.float32), rest[:, 0, 1].numpy())
    for k, ref in (("xyz", xyz), ("features_dc", dc), ("features_rest", rest), ("opacity", op), ("scaling", sc), ("rotation", rot)):
        assert out[k].shape == ref.shape and torch.equal(out[k], ref), k


def test_anchor_round_trip_scaffold_and_octree():
    g = torch.Generator().manual_seed(4)
    N, k = 9, 10
    anchor = torch.randn(N, 3, generator=g); off = torch.randn(N, k, 3, generator=g); feat = torch.randn(N, 32, generator=g)
    op = torch.randn(N, 1, generator=g); sc = torch.randn(N, 6, generator=g); rot = torch.randn(N, 4, generator=g)
    level = torch.randint(0, 5, (N, 1), generator=g); extra = torch.rand(N, generator=g)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "s.ply")
        ply.save_anchors(p, anchor, off, feat, op, sc, rot)
        names, a = ply.read_vertex_table(p)
        out = ply.load_anchors(p)
        assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[6] == "f_offset_0" and names[6 + 3 * k] == "f_anchor_feat_0"
        assert np.array_equal(a[:, 6 + 2 * k + 3].astype(np.float32), off[:, 3, 2].numpy())         # offsets channel-major like f_rest
        assert "level" not in out
        for key, ref in (("anchor", anchor), ("offset", off), ("anchor_feat", feat), ("opacity", op), ("scaling", sc), ("rotation", rot)):
            assert torch.equal(out[key], ref), key
        q = os.path.join(d, "o.ply")
        ply.save_anchors(q, anchor, off, feat, op, sc, rot, level=level, extra_level=extra, voxel_size=0.02, standard_dist=7.5)
        names, a = ply.read_vertex_table(q)
        out = ply.load_anchors(q)
        assert names[6:9] == ["level", "extra_level", "info"] and names[9] == "f_offset_0"
        assert torch.equal(out["level"], level.to(torch.int32)) and torch.equal(out["extra_level"], extra) and torch.equal(out["offset"], off)
        assert out["voxel_size"] == np.float32(0.02) and out["standard_dist"] == 7.5 and np.all(a[2:, 8] == 0)
        with pytest.raises(ValueError):
            ply.save_anchors(q, anchor, off, feat, op, sc, rot, level=level)


def test_layout_matches_the_reference_run():
    """tests/golden/ref_ply_layout.npz (make_golden_ref.py: the reference's own save_gaussians / load_gaussians with plyfile's describe / read
    probed): our files carry the same column names and values, and loading the reference's table gives what the reference loaded."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_ply_layout.npz"))
    T = lambda k: torch.tensor(z[k])
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "v.ply")
        ply.save_gaussians(p, T("vanilla_in_xyz"), T("vanilla_in_features_dc"), T("vanilla_in_features_rest"), T("vanilla_in_opacity"),
                           T("vanilla_in_scaling"), T("vanilla_in_rotation"))
        names, a = ply.read_vertex_table(p)
        assert names == list(z["vanilla_names"]) and np.array_equal(a.astype(np.float32), z["vanilla_table"])
        ply.write_vertex_table(p, list(z["vanilla_names"]), z["vanilla_table"])           # the reference's file -> our loader
        out = ply.load_gaussians(p, 3)
        for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
            assert np.array_equal(out[k].numpy(), z["vanilla_loaded_" + k]), k
        for tag in ("scaffold", "octree"):
            oc = tag == "octree"
            kw = dict(level=T("octree_in_level"), extra_level=T("octree_in_extra_level"), voxel_size=float(z["octree_loadedvoxel_size"]),
                      standard_dist=float(z["octree_loadedstandard_dist"])) if oc else {}
            ply.save_anchors(p, T(tag + "_in_anchor"), T(tag + "_in_offset"), T(tag + "_in_anchor_feat"), T(tag + "_in_opacity"), T(tag + "_in_scaling"),
                             T(tag + "_in_rotation"), **kw)
            names, a = ply.read_vertex_table(p)
            assert names == list(z[tag + "_names"]) and np.array_equal(a.astype(np.float32), z[tag + "_table"]), tag
            ply.write_vertex_table(p, list(z[tag + "_names"]), z[tag + "_table"])
            out = ply.load_anchors(p)
            for k in ("anchor", "offset", "anchor_feat", "opacity", "scaling", "rotation") + (("level", "extra_level") if oc else ()):
                assert np.array_equal(out[k].numpy(), z[f"{tag}_loaded_{k}"]), (tag, k)
            if oc:
                assert out["voxel_size"] == float(z["octree_loadedvoxel_size"]) and out["standard_dist"] == float(z["octree_loadedstandard_dist"])
