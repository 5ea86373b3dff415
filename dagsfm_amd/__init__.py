"""dagsfm_amd -- MI355X-native matching + two-view verification path of DAGSfM.

The product is the C-ABI shared library `libdagsfm_mi355x.so` (HIP kernels for gfx950 under
`csrc/`, declared in `include/dagsfm_mi355x.h`).  This package only binds it for the Python
tests and `bench.py`; there is no CPU fallback -- importing `capi` without the built library, or
creating a context without a GPU, raises.
"""
__all__ = ["capi", "synthetic"]
