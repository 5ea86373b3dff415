"""CPU-only: the C-ABI library builds, loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms.update(re.findall(r"\b(dsm_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    from dagsfm_amd import capi
    assert os.path.exists(capi.LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s


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
    import torch
    from dagsfm_amd import capi
    if torch.cuda.is_available():
        return
    try:
        capi.Context(0)
    except capi.DsmError as e:
        assert "dsm_ctx_create failed" in str(e)
    else:
        raise AssertionError("context creation must fail without a HIP device")
