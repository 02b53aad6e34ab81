"""The C-ABI library loads and exports every symbol include/f3dg.h declares (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "f3dg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(f3dg_\w+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound(f3d):
    from f3dgaus_amd import _lib
    names = _declared()
    assert len(names) >= 12
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"libf3dg_hip.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), (set(names) ^ set(_lib.SIGNATURES))
    assert _lib.lib().f3dg_version().startswith(b"f3dg-hip gfx950")


def test_library_is_gfx950_code_object():
    lib = os.path.join(ROOT, "f3d-gaus_amd", "csrc", "libf3dg_hip.so")
    data = open(lib, "rb").read()
    assert b"gfx950" in data and b"render3s_fwd_kernel" in data


def test_workspace_arithmetic_and_host_side_errors(f3d):
    from f3dgaus_amd import _lib
    L = _lib.lib()
    a = L.f3dg_workspace_bytes(65536, 256, 256, 1, 200000)
    b = L.f3dg_workspace_bytes(65536, 256, 256, 1, 400000)
    c = L.f3dg_workspace_bytes(65536, 256, 256, 8, 400000)
    assert 0 < a < b < c
    assert b - a >= 200000 * 20                       # per instance: two u32 id halves, a 4-byte group half and the 8-byte half that
                                                      # also holds the exported 64-bit keys
    assert L.f3dg_workspace_bytes(-1, 256, 256, 1, 10) == 0
    assert L.f3dg_workspace_bytes(10, 0, 256, 1, 10) == 0
    # argument validation happens before any HIP call, so it is testable without a GPU
    null = None
    rc = L.f3dg_forward_batched(null, null, 0, 10, 1, 10, 1, 4, null, 64, 64, null, null, null, null, null, 1.0,
                                null, null, null, null, null, null, 0.1, 0.1, 0.0, null, null, 0)
    assert rc == _lib.ERR_BAD_ARG
    buf = (C.c_char * 1024)()
    one = C.cast(buf, C.c_void_p)
    rc = L.f3dg_forward_batched(null, one, 1024, 100000, 1, 1000, 1, 4, one, 64, 64, one, one, null, one, one, 1.0,
                                one, null, null, one, one, one, 0.1, 0.1, 0.0, one, null, 0)
    assert rc == _lib.ERR_WORKSPACE
    assert L.f3dg_splat_head(null, 0, 32, 32, *([null] * 5), 1.0, 1024, 0, *([null] * 7)) == _lib.ERR_BAD_ARG
    assert L.f3dg_mark_visible(null, 5, null, null, null, null) == _lib.ERR_BAD_ARG


def test_python_api_errors_without_gpu(f3d):
    import torch
    S = f3d.GaussianRasterizationSettings_GOF
    assert S._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "kernel_size", "subpixel_offset", "bg",
                         "scale_modifier", "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    rs = S(64, 64, 0.1, 0.1, 0.0, torch.zeros(0), torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 1, torch.zeros(3),
           False, False)
    r = f3d.GaussianRasterizer_GOF(rs)
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 4, 3))
    with pytest.raises(RuntimeError, match="HIP device"):       # no silent CPU fallback
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 4, 3), scales=x,
          rotations=torch.zeros(4, 4))


def test_missing_extension_fails_loudly(f3d, monkeypatch, tmp_path):
    """No silent CPU / PyTorch fallback: without the built HIP library the binding raises."""
    from f3dgaus_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libf3dg_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _lib.lib()


def test_product_package_never_imports_the_oracle():
    import re
    pkg = os.path.join(ROOT, "f3d-gaus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)
                assert "libgof_oracle" not in txt, os.path.join(dirpath, f)


def test_round3_host_logic(f3d):
    """Options, diagnostics, the small-call path's workspace carving and the integrate layouts: pure host arithmetic."""
    from f3dgaus_amd import _lib
    L = _lib.lib()
    # the default library knows fifteen options and two diagnostics; the switches of the superseded generations exist in lab builds only
    defaults = ((b"render_fast", 1), (b"tile_cull", 1), (b"small_path", 2), (b"small_path_aux", 1), (b"render_lowocc", 1), (b"render_split", -1),
                (b"render_unroll", -1), (b"render_pack", -1), (b"render_pack_th", 32), (b"render_scan", -1), (b"render_scan_th", 12), (b"render_scan_min", 4), (b"bwd_dense", 1), (b"bwd_occ", 5),
                (b"render_count", 0), (b"time_launches", 0))
    for name, v in defaults:
        assert L.f3dg_set_option(name, v) == 0, name
    lab_only = ((b"render_kernel", 3), (b"render_slide", 1), (b"render_dma", 1), (b"render_lds_pad", 0), (b"small_debug", 0), (b"render_tail", 16), (b"render_tail", -1),
                (b"sort_fused_rects", 1), (b"sort_fused_rects", 0), (b"pre_order", 3), (b"pre_order", 0), (b"render_wpb", 1), (b"render_pretest", 1),
                (b"render_cull", 1), (b"render_queue", 1), (b"render_round", 192), (b"sort_wide_groups", 0), (b"pre_hoist", 0), (b"debug_skip_all", 0))
    lab = L.f3dg_version().endswith(b"lab")
    for name, v in lab_only:
        assert L.f3dg_set_option(name, v) == (0 if lab else _lib.ERR_BAD_ARG), name
    assert L.f3dg_set_option(b"no_such_option", 1) == _lib.ERR_BAD_ARG
    assert L.f3dg_debug_launch_count(1) >= 0 and L.f3dg_debug_launch_count(0) == 0
    # the per-tile slots of the small-call path exist for one or two views of at most 2^18 Gaussians only: 4096 x 4 B per (view, tile)
    T = 256
    two = L.f3dg_workspace_bytes(65536, 256, 256, 2, 100000)
    three = L.f3dg_workspace_bytes(65536, 256, 256, 3, 100000)
    one = L.f3dg_workspace_bytes(65536, 256, 256, 1, 100000)
    assert two - one > T * 4096 * 4                  # a second view's slots on top of its per-view arrays
    assert three < two + (two - one) - T * 4096 * 4 + 4096      # the third view brings no slots (and frees those of the other two)
    assert L.f3dg_workspace_bytes((1 << 18) + 1, 256, 256, 1, 100000) - L.f3dg_workspace_bytes(1 << 18, 256, 256, 1, 100000) < 1 << 20
    # batched integrate workspaces grow by the per-camera tables (2 KB of contributor ids per pixel)
    i1 = L.f3dg_integrate_workspace_bytes_batched(1000, 100, 64, 64, 1, 50000)
    i4 = L.f3dg_integrate_workspace_bytes_batched(1000, 100, 64, 64, 4, 50000)
    assert i1 == L.f3dg_integrate_workspace_bytes(1000, 100, 64, 64, 50000) and i4 - i1 >= 3 * 64 * 64 * 2048
    assert L.f3dg_integrate_workspace_bytes_batched(1000, 100, 64, 64, 0, 50000) == 0
    # more than 2^28 Gaussians do not fit a list entry (28-bit id + quadrant mask)
    buf = (C.c_char * 1024)()
    one_p = C.cast(buf, C.c_void_p)
    rc = L.f3dg_forward_batched(None, one_p, 1024, 1000, 1, (1 << 28) + 5, 1, 4, one_p, 64, 64, one_p, one_p, None, one_p, one_p, 1.0,
                                one_p, None, None, one_p, one_p, one_p, 0.1, 0.1, 0.0, one_p, None, 0)
    assert rc == _lib.ERR_BAD_ARG
    # several Gaussian sets are an inference path
    rc = L.f3dg_forward_sets(None, one_p, 1 << 40, 1000, 2, 1, 100, 1, 4, one_p, 64, 64, one_p, one_p, None, one_p, one_p, 1.0,
                             one_p, None, None, one_p, one_p, one_p, 0.1, 0.1, 0.0, one_p, None, _lib.FLAG_SAVE_AUX)
    assert rc == _lib.ERR_BAD_ARG


def test_stream_cache_and_alias_modules(f3d):
    from f3dgaus_amd.diff_gof_rasterization import _StreamCache
    c = _StreamCache(limit=3)
    for i in range(5):
        c[i] = i
    assert list(c) == [2, 3, 4] and c.get(2) == 2
    c[9] = 9
    assert list(c) == [4, 2, 9] and c.get(77) is None
    import f3dgaus_amd.diff_gof_rasterization.backward as b
    assert b.__spec__.name == "f3d-gaus_amd.diff_gof_rasterization.backward" and b.__package__ == "f3d-gaus_amd.diff_gof_rasterization"


def test_flag_constants_match_the_header():
    """The flag bits of f3dg_forward_sets as Python sees them are the header's."""
    import re
    from f3dgaus_amd import _lib
    text = open(HEADER).read() if "HEADER" in globals() else open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "f3dg.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define F3DG_FLAG_(\w+)\s+(\d+)u", text)}
    for name in ("SAVE_AUX", "BG_PER_VIEW", "SKIP_NORMAL", "SKIP_DISTORTION", "EXACT", "FAST", "NO_TILE_CULL", "NO_SMALL_PATH"):
        assert getattr(_lib, "FLAG_" + name) == defs[name], name
    assert len(set(defs.values())) == len(defs)


def test_bench_pmc_counter_mean_takes_the_largest_grid_dispatches(tmp_path):
    """bench.pmc_counter_mean (the parser behind roofline.traffic): mean over the dispatches of the named kernel with the largest
    grid, other kernels / counters / smaller set-up launches ignored; None when nothing matches."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    d = tmp_path / "pass" / "host"
    d.mkdir(parents=True)
    hdr = "Dispatch_Id,Kernel_Name,Grid_Size,Counter_Name,Counter_Value\n"
    rows = [(1, "void (anonymous namespace)::render3s_fwd_kernel<false, true>(int)", 64 * 100, "FETCH_SIZE", 5.0),      # set-up launch
            (2, "void (anonymous namespace)::render3s_fwd_kernel<false, true>(int)", 64 * 4000, "FETCH_SIZE", 100.0),
            (3, "void (anonymous namespace)::render3s_fwd_kernel<false, true>(int)", 64 * 4000, "FETCH_SIZE", 110.0),
            (4, "void (anonymous namespace)::render3s_fwd_kernel<false, true>(int)", 64 * 4000, "WRITE_SIZE", 7.0),
            (5, "void (anonymous namespace)::preprocess_kernel<false, false>(int)", 64 * 9000, "FETCH_SIZE", 900.0)]
    (d / "b_counter_collection.csv").write_text(hdr + "".join('%d,"%s",%d,%s,%r\n' % r for r in rows))
    assert bench.pmc_counter_mean(str(tmp_path), "render3s_fwd_kernel", "FETCH_SIZE") == (105.0, 2)
    assert bench.pmc_counter_mean(str(tmp_path), "render3s_fwd_kernel", "WRITE_SIZE") == (7.0, 1)
    assert bench.pmc_counter_mean(str(tmp_path), "render5_bwd_kernel", "FETCH_SIZE") is None
