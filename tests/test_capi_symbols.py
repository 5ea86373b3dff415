"""CPU-only: the C-ABI library builds, loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsm_[a-z0-9_]+)\s*\(", text)))


def test_every_header_is_covered():
    assert sorted(os.path.basename(h) for h in glob.glob(os.path.join(ROOT, "include", "*.h"))) == ["dagsfm_gather.h", "dagsfm_mi355x.h"]


def test_library_exports_every_declared_symbol():
    from dagsfm_amd import capi
    assert os.path.exists(capi.LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    syms = declared_symbols("dagsfm_mi355x.h")
    assert len(syms) >= 10
    for path in (capi.LIB_PATH, capi.CHECK_LIB_PATH):  # the product and the check build of the same sources
        assert os.path.exists(path), path
        lib = ctypes.CDLL(path)
        for s in syms:
            assert hasattr(lib, s), "missing export in %s: %s" % (os.path.basename(path), s)


def test_gather_library_exports_every_declared_symbol():
    """include/dagsfm_gather.h = libdagsfm_gather.so, the RCCL companion (maps librccl: a library of its own for that reason)."""
    from dagsfm_amd import capi
    assert os.path.exists(capi.GATHER_LIB_PATH), capi.GATHER_LIB_PATH
    lib = ctypes.CDLL(capi.GATHER_LIB_PATH)
    syms = [s for s in declared_symbols("dagsfm_gather.h") if s.startswith("dsm_gather_")]
    assert len(syms) == 8, syms
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    import subprocess
    needed = subprocess.run(["ldd", capi.GATHER_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "librccl" in needed and "libdagsfm_mi355x.so" in needed
    product = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "librccl" not in product, "a single-GPU host must not map RCCL"


def test_product_library_is_lean():
    """the cross-check schedules live in the check build; the product knows at most 8 scheduling knobs, its
    kernels do not include the legacy / comparison ones, and it stays under 3.5 MB."""
    import subprocess
    from dagsfm_amd import capi
    assert len(capi.PRODUCT_OPTION_KEYS) <= 8 and not set(capi.PRODUCT_OPTION_KEYS) & set(capi.CHECK_OPTION_KEYS)
    assert os.path.getsize(capi.LIB_PATH) <= 3.5 * 2 ** 20
    ctx_h = open(os.path.join(ROOT, "dagsfm_amd", "csrc", "ctx.h")).read()
    for key in capi.PRODUCT_OPTION_KEYS + capi.CHECK_OPTION_KEYS:  # the binding's key lists are the library's
        assert '"%s"' % key in ctx_h, key
    names = {}
    for path in (capi.LIB_PATH, capi.CHECK_LIB_PATH):
        names[path] = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    for kernel in ("k_ransac", "k1_best_rows_dot4", "k_roots_e_lds"):
        assert kernel not in names[capi.LIB_PATH], kernel
        assert kernel in names[capi.CHECK_LIB_PATH], kernel


def test_option_defaults_match_reference():
    # /root/reference/src/feature/sift.h:116-165, two_view_geometry.h:105-143
    from dagsfm_amd import capi
    m = capi.default_match_options()
    assert (m.max_ratio, m.max_distance, m.cross_check, m.max_num_matches) == (0.8, 0.7, 1, 32768)
    t = capi.default_two_view_options()
    assert (t.min_num_inliers, t.min_E_F_inlier_ratio, t.max_H_inlier_ratio) == (15, 0.95, 0.8)
    assert (t.watermark_min_inlier_ratio, t.watermark_border_size, t.detect_watermark) == (0.7, 0.1, 1)
    assert (t.max_error, t.min_inlier_ratio, t.confidence, t.min_num_trials, t.max_num_trials) == \
        (4.0, 0.25, 0.999, 30, 10000)


def test_no_device_fails_loudly():
    """Without a GPU the product must refuse to run (no CPU fallback)."""
    import pytest
    import torch
    from dagsfm_amd import capi
    # (/dev/kfd: a process whose HIP runtime was brought up by this library before torch asked can report no torch device on a GPU box)
    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    try:
        capi.Context(0)
    except capi.DsmError as e:
        assert "dsm_ctx_create failed" in str(e)
    else:
        raise AssertionError("context creation must fail without a HIP device")
