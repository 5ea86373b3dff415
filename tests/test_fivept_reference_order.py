"""The oracle's 5-point steps 3 / 4 (fivept_build_A, fivept_det_poly: expanded from the term table
dagsfm_amd/csrc/fivept_terms.tbl) against the reference's OWN generated headers
(/root/reference/src/estimators/essential_matrix_poly.h, essential_matrix_coeffs.h) compiled behind a shim into
oracle/_ref/libfivept_ref.so -- bit for bit.  The _ref library exists where the oracle was built next to
/root/reference (this container; it travels with the tree); without it the test is skipped."""
import ctypes
import os

import numpy as np
import pytest

from tests import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libfivept_ref.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference at build time)")

DP = ctypes.POINTER(ctypes.c_double)


def _fn(lib, name):
    f = getattr(lib, name)
    f.argtypes = [DP, DP]
    f.restype = None
    return f


def _call(f, x, nout):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.full(nout, np.nan)
    f(x.ctypes.data_as(DP), out.ctypes.data_as(DP))
    return out


def _inputs(rng, n, trial):
    kind = trial % 4
    if kind == 0:
        return rng.standard_normal(n)
    if kind == 1:   # a null-space basis: orthonormal columns
        q, _ = np.linalg.qr(rng.standard_normal((9, 4)))
        return np.resize(q.T.ravel(), n)
    if kind == 2:   # wide dynamic range
        return rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)
    return np.round(rng.standard_normal(n) * 4) / 4   # many exact cancellations


@needs_ref
def test_constraint_matrix_and_determinant_polynomial_bit_exact():
    ref = ctypes.CDLL(REF_SO)
    orc = oracle_lib.load().lib
    rng = np.random.default_rng(5)
    for trial in range(2000):
        e = _inputs(rng, 36, trial)
        a_ref = _call(_fn(ref, "ref_fivept_build_A"), e, 200)
        a_orc = _call(_fn(orc, "oracle_fivept_build_A"), e, 200)
        assert not np.isnan(a_ref).any()
        assert a_ref.tobytes() == a_orc.tobytes()
        b = _inputs(rng, 39, trial)
        c_ref = _call(_fn(ref, "ref_fivept_coeffs"), b, 11)
        c_orc = _call(_fn(orc, "oracle_fivept_coeffs"), b, 11)
        assert not np.isnan(c_ref).any()
        assert c_ref.tobytes() == c_orc.tobytes()


def test_table_covers_every_entry_once():
    tbl = os.path.join(ROOT, "dagsfm_amd", "csrc", "fivept_terms.tbl")
    rows = [l.split()[:2] for l in open(tbl) if l[0] in "AC"]
    assert sorted(int(i) for k, i in rows if k == "A") == list(range(200))
    assert sorted(int(i) for k, i in rows if k == "C") == list(range(11))


def test_constraint_matrix_is_nisters_system():
    """Independent of the table: for E = x E0 + y E1 + z E2 + E3 the rows of A are det(E) and
    E E^T E - 0.5 trace(E E^T) E expanded in the 20 cubic monomials -- checked by evaluating A . m(x, y, z)."""
    orc = oracle_lib.load().lib
    rng = np.random.default_rng(11)
    e = rng.standard_normal(36)
    a = _call(_fn(orc, "oracle_fivept_build_A"), e, 200).reshape(20, 10).T   # A(r, c)
    Eb = e.reshape(4, 9)                                                      # column c of the 9 x 4 basis
    worst = 0.0
    for _ in range(20):
        x, y, z = rng.standard_normal(3)
        E = (x * Eb[0] + y * Eb[1] + z * Eb[2] + Eb[3]).reshape(3, 3)
        cons = np.concatenate([[np.linalg.det(E)], (E @ E.T @ E - 0.5 * np.trace(E @ E.T) * E).ravel()])
        mono = np.array([x**3, y**3, x*x*y, x*y*y, x*x*z, x*x, y*y*z, y*y, x*y*z, x*y,
                         x*z*z, x*z, x, y*z*z, y*z, y, z**3, z*z, z, 1.0])
        got = a @ mono
        # rows may be ordered / signed differently from this test's listing: compare as sets of |values|
        worst = max(worst, np.abs(np.sort(np.abs(got)) - np.sort(np.abs(cons))).max())
    assert worst < 1e-9
