"""dagsfm_amd/csrc/exact_trig.h: the atan / sin / cos / tan of the five trigonometric camera models, computed in
double-double arithmetic so that the device and the oracle return the same -- the correctly rounded -- bits.

CPU: the host build (inside the oracle) against mpmath's correctly rounded values (exact agreement) and against this
host's libm, i.e. what a build of the reference would call (never more than 1 ulp apart, equal for > 99 % of the
arguments: glibc is not correctly rounded, the difference is glibc's).  GPU: the device build against the host build
through Camera::ImageToWorld, bit for bit (tests/test_camera_models.py)."""
import ctypes
import math

import numpy as np
import pytest

DP = ctypes.POINTER(ctypes.c_double)
NAMES = ["atan", "sin", "cos", "tan"]
LIBM = [math.atan, math.sin, math.cos, math.tan]


def _run(oracle, kind, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    oracle.lib.oracle_exact_trig(kind, x.ctypes.data_as(DP), len(x), out.ctypes.data_as(DP))
    return out


def _args(kind, rng, n):
    if kind == 0:
        return np.concatenate([rng.uniform(0, 4, n), 10.0 ** rng.uniform(-12, 12, n // 2), -rng.uniform(0, 30, n // 4),
                               [1.0, 0.125, 0.5, 0.0625, 1e-9, 1e19, 2.0 ** 53, 1e300]])
    return np.concatenate([rng.uniform(0, 3.2, n), rng.uniform(-20, 20, n // 2), 10.0 ** rng.uniform(-12, 0, n // 4),
                           rng.uniform(0, 1e5, n // 4), [math.pi / 2, math.pi, 3 * math.pi / 2, math.pi / 4, 1e-300, 1647099.0, -1647099.0]])


@pytest.mark.parametrize("kind", range(4))
def test_correctly_rounded(oracle, kind):
    mpmath = pytest.importorskip("mpmath")
    mpmath.mp.prec = 300
    fn = [mpmath.atan, mpmath.sin, mpmath.cos, mpmath.tan][kind]
    xs = _args(kind, np.random.default_rng(100 + kind), 8000)
    got = _run(oracle, kind, xs)
    for x, g in zip(xs, got):
        assert g == float(fn(mpmath.mpf(float(x)))), (NAMES[kind], float(x))


@pytest.mark.parametrize("kind", range(4))
def test_within_one_ulp_of_the_host_libm(oracle, kind):
    xs = _args(kind, np.random.default_rng(200 + kind), 100000)
    got = _run(oracle, kind, xs)
    ref = np.array([LIBM[kind](float(v)) for v in xs])
    ulp = np.abs(got - ref) / np.spacing(np.abs(ref))
    assert ulp.max() <= 1.0, (NAMES[kind], xs[np.argmax(ulp)])
    assert (got == ref).mean() > 0.99


def test_special_values(oracle):
    assert list(_run(oracle, 0, [0.0, -0.0, np.inf, -np.inf])) == [0.0, -0.0, math.pi / 2, -math.pi / 2]
    assert math.copysign(1.0, _run(oracle, 0, [-0.0])[0]) == -1.0
    assert list(_run(oracle, 1, [0.0])) == [0.0] and list(_run(oracle, 2, [0.0])) == [1.0] and list(_run(oracle, 3, [0.0])) == [0.0]
    for kind in range(1, 4):
        assert np.isnan(_run(oracle, kind, [np.inf, np.nan])).all()
    assert np.isnan(_run(oracle, 0, [np.nan])).all()


def test_arguments_beyond_the_supported_range_are_nan(oracle):
    # |x| <= 1 647 099 (< 2^20 pi/2) is the range of the exact argument reduction; beyond it sin / cos / tan are NaN in the
    # oracle and on the device alike (the integer cast of the quadrant is undefined from ~1.4e19 on)
    for kind in range(1, 4):
        assert np.isnan(_run(oracle, kind, [1647099.5, -1647100.0, 1e7, 1.4e19, -1e300, 1.7e308])).all()
        assert np.isfinite(_run(oracle, kind, [1647099.0, -1647099.0])).all()
    assert np.isfinite(_run(oracle, 0, [1e19, 1e300, -1.7e308])).all()  # atan has no reduction by pi/2
