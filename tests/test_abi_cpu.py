"""CPU-side checks of the product boundary (no GPU, no compute calls): the C-ABI library loads, exports every symbol
include/gsrast.h declares, sizes its arenas sanely, and the python drop-in packages keep the reference's names,
signatures and error behaviour."""
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gsrast.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import gsrast
    L = gsrast.lib()
    decl = _declared_symbols()
    assert len(decl) >= 15
    for s in decl:
        assert hasattr(L, s), f"libgsrast_hip.so does not export {s}"
    assert sorted(gsrast.EXPORTS) == decl
    assert L.gsr_abi_version() == gsrast.ABI_VERSION


def test_decode_exports_match_header():
    import gsrast
    from gsrast import decode
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gsdecode.h")).read(), flags=re.S)
    decl = sorted(set(re.findall(r"\b(gsd_[a-z0-9_]+)\s*\(", src)))
    assert sorted(decode.EXPORTS) == decl and len(decl) == 13
    L = gsrast.lib()
    for s in decl:
        assert hasattr(L, s), f"libgsrast_hip.so does not export {s}"
    Ld = decode._lib()
    c = decode.Cfg(100000, 60000, 10, 32, 0, 0, 0, 0)
    b = Ld.gsd_backward_scratch_bytes(__import__("ctypes").byref(c))
    assert 60000 * 416 * 4 <= b <= 60000 * 416 * 4 + (16 << 20)        # feature-major columns + partial tiles
    assert Ld.gsd_forward_scratch_bytes(60000) < (1 << 17)           # weight image (101 KB) + scan temporaries
    with pytest.raises(RuntimeError, match="mlp_feature_bank"):      # the feature-bank branch needs its MLP (get_featurebank_mlp)
        decode.neural_gaussians(torch.zeros(2, 3), torch.zeros(2, 32), torch.zeros(2, 10, 3), torch.zeros(2, 6), None, None, None,
                                torch.zeros(3), use_feat_bank=True)


def test_arena_sizes_scale_linearly():
    import gsrast
    L = gsrast.lib()
    for v in (gsrast.EWA, gsrast.SURFEL, gsrast.PLANE):
        g1, g2 = L.gsr_geom_bytes(v, 100000), L.gsr_geom_bytes(v, 200000)
        assert 1.8 < g2 / g1 < 2.2 and g1 / 100000 < 200          # < 200 B per gaussian of private state
        b = L.gsr_binning_bytes(v, 3000000, 1920, 1080)
        table = (256 + 1) * 8160 * 4 + 256 * 256 * 4               # bucket sort on the tile id: <= 256 chunk rows of per-tile counts (+ totals), chunk x group prefixes
        assert 16 * 3000000 <= b <= 20 * 3000000 + table           # 4 x u32 per instance (+ histograms) + that table
        i = L.gsr_img_bytes(v, 1920, 1080)
        assert i >= 1920 * 1080 * 8
    assert L.gsr_img_bytes(gsrast.SURFEL, 1920, 1080) > L.gsr_img_bytes(gsrast.EWA, 1920, 1080)


def test_dropin_signatures_match_reference():
    import diff_gaussian_rasterization as dgr
    import diff_surfel_rasterization as dsr
    import diff_plane_rasterization as dpr
    import scaffold_filter as sf
    base = ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
            "sh_degree", "campos", "prefiltered", "debug"]
    assert list(dgr.GaussianRasterizationSettings._fields) == base
    assert list(dsr.GaussianRasterizationSettings._fields) == base
    assert list(sf.GaussianRasterizationSettings._fields) == base
    assert list(dpr.GaussianRasterizationSettings._fields) == base[:-1] + ["render_geo", "debug"]
    fwd = ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"]
    assert list(inspect.signature(dgr.GaussianRasterizer.forward).parameters) == fwd
    assert list(inspect.signature(dsr.GaussianRasterizer.forward).parameters) == fwd
    assert list(inspect.signature(dpr.GaussianRasterizer.forward).parameters) == \
        ["self", "means3D", "means2D", "means2D_abs", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp", "all_map"]
    assert list(inspect.signature(sf.GaussianRasterizer.visible_filter).parameters) == ["self", "means3D", "scales", "rotations", "cov3D_precomp"]
    assert hasattr(dgr.GaussianRasterizer, "markVisible") and hasattr(dgr, "rasterize_gaussians")
    from simple_knn._C import distCUDA2
    assert callable(distCUDA2)


def _settings(mod, **kw):
    z = torch.zeros(3)
    base = dict(image_height=32, image_width=32, tanfovx=0.5, tanfovy=0.5, bg=z, scale_modifier=1.0, viewmatrix=torch.eye(4),
                projmatrix=torch.eye(4), sh_degree=0, campos=z, prefiltered=False, debug=False)
    base.update(kw)
    return mod.GaussianRasterizationSettings(**base)


def test_argument_validation_messages():
    import diff_gaussian_rasterization as dgr
    import diff_plane_rasterization as dpr
    r = dgr.GaussianRasterizer(_settings(dgr))
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, torch.zeros(4, 1), scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, torch.zeros(4, 1), shs=torch.zeros(4, 16, 3), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, torch.zeros(4, 1), colors_precomp=m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 6))
    rp = dpr.GaussianRasterizer(_settings(dpr, render_geo=True))
    with pytest.raises(Exception, match="excatly one of either SHs"):
        rp(m, m, m, torch.zeros(4, 1), scales=m, rotations=torch.zeros(4, 4))


def test_no_cpu_fallback():
    """The product must fail loudly for host tensors (the reference is CUDA-only too); it never routes to the oracle."""
    import diff_gaussian_rasterization as dgr
    r = dgr.GaussianRasterizer(_settings(dgr))
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        r(m, m, torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="dimensions"):
        r(torch.zeros(4, 2), m, torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))
    # product sources never reference the oracle
    for dp, _, fs in os.walk(os.path.join(ROOT, "gs-sr_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                for pat in ("import oracle", "from oracle", "gsr_oracle", "libgsr_oracle", "oracle/"):
                    assert pat not in txt, (os.path.join(dp, f), pat)


def test_decode_argument_errors_without_a_device():
    """gsd_* validate their arguments before touching the device: messages are checkable on a CPU-only box."""
    import ctypes as C
    import gsrast
    from gsrast import decode
    L = decode._lib()
    buf = (C.c_float * 64)()
    addr = C.addressof(buf)
    prm = decode.Params(*[addr] * 13)
    inp = decode.Inputs(*[addr] * 8)
    P = C.c_uint32(0)

    def stage1(cfg):
        return L.gsd_forward_stage1(C.byref(cfg), C.byref(inp), C.byref(prm), addr, addr, addr, C.byref(P), addr, 1 << 20, None)
    assert stage1(decode.Cfg(10, 5, 17, 0, 0, 0, 0, 0)) != 0 and "n_offsets" in gsrast.last_error()
    assert stage1(decode.Cfg(10, 5, 10, 65, 0, 0, 0, 0)) != 0 and "appearance_dim" in gsrast.last_error()
    assert stage1(decode.Cfg(10, 11, 10, 0, 0, 0, 0, 0)) != 0 and "bad sizes" in gsrast.last_error()
    bad = decode.Inputs(addr, addr + 4, addr, addr, None, None, addr, addr)              # feat not 16-byte aligned
    assert L.gsd_forward_stage1(C.byref(decode.Cfg(10, 5, 10, 0, 0, 0, 0, 0)), C.byref(bad), C.byref(prm), addr, addr, addr, C.byref(P), addr,
                                1 << 20, None) != 0 and "aligned" in gsrast.last_error()
    lvl = decode.Inputs(addr, addr, addr, addr, None, None, addr, addr)
    assert L.gsd_forward_stage1(C.byref(decode.Cfg(10, 5, 10, 0, 0, 0, 0, 1)), C.byref(lvl), C.byref(prm), addr, addr, addr, C.byref(P), addr,
                                1 << 20, None) != 0 and "level" in gsrast.last_error()
    assert L.gsd_forward_stage1(C.byref(decode.Cfg(10, 5, 10, 0, 0, 0, 0, 0)), C.byref(inp), C.byref(prm), addr, addr, addr, C.byref(P), addr,
                                16, None) != 0 and "scratch" in gsrast.last_error()


def test_multiview_loss_argument_errors_without_a_device():
    """gsr_loss_plane_mv_* reject bad configurations / null pointers / short scratch before launching anything."""
    import ctypes as C
    import gsrast
    L = gsrast.lib()
    buf = (C.c_float * 64)()
    a = C.addressof(buf)
    good = gsrast.MvCfg(8, 4, 8, 4, 8, 4, 10, 10, 4, 2, 10, 10, 4, 2)
    good.ncc_scale, good.noise_th, good.patch = 1.0, 1.0, 3
    bad = gsrast.MvCfg(0, 4, 8, 4, 8, 4, 10, 10, 4, 2, 10, 10, 4, 2)
    bad.ncc_scale, bad.patch = 1.0, 3
    big = 1 << 16
    assert L.gsr_loss_plane_mv_geo(C.byref(bad), a, a, a, a, a, a, a, a, a, big, None) != 0 and "configuration" in gsrast.last_error()
    assert L.gsr_loss_plane_mv_geo(C.byref(good), a, None, a, a, a, a, a, a, a, big, None) != 0 and "null pointer" in gsrast.last_error()
    assert L.gsr_loss_plane_mv_geo(C.byref(good), a, a, a, a, a, a, a, a, a, 0, None) != 0 and "scratch" in gsrast.last_error()
    good.patch = 9
    assert L.gsr_loss_plane_mv_ncc(C.byref(good), 4, a, a, a, a, a, a, None, None, a, a, a, a, big, None) != 0 and "configuration" in gsrast.last_error()
    good.patch = 3
    assert L.gsr_loss_plane_mv_ncc(C.byref(good), 4, None, a, a, a, a, a, None, None, a, a, a, a, big, None) != 0 and "null pointer" in gsrast.last_error()
    assert L.gsr_loss_plane_mv_scratch_bytes(1920, 1080, 102400) >= 8 * max(60 * 135, 6400)
