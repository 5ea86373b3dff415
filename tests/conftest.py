import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()


class _BuildSwitch:
    """What the `dsm` fixture hands out: the PRODUCT library's context -- or, while the test has a cross-check switch in the
    environment (monkeypatch.setenv("DSM_VERIFY_LEGACY", ...), DSM_SCORE_PREFILTER, DSM_K1_DOT4 ...: capi.CHECK_OPTION_KEYS), a
    context of the check build (libdagsfm_mi355x_check.so), where those schedules exist.  A test sets its switches before its
    first call, so every call of a test lands on the same context; the product context never sees a check-only key (it would
    refuse it: dsm_set_debug_option fails on an unknown key)."""

    def __init__(self):
        self._ctx = {}

    def _pick(self):
        from dagsfm_amd import capi
        check = capi.check_requested()
        if check not in self._ctx:
            self._ctx[check] = capi.Context(0, check=check)
        return self._ctx[check]

    def __getattr__(self, name):
        return getattr(self._pick(), name)


@pytest.fixture(scope="session")
def dsm():
    """The product C-ABI on HIP device 0 (the check build while a cross-check switch is set).  Fails loudly when the library or
    the GPU is missing."""
    return _BuildSwitch()
