"""Runs the CPU-only tests against AddressSanitizer + UBSan builds (tools/sanitize.sh sets the environment): the oracle as a whole,
and the host-only part of the product library (EKF, formats, caller glue -- the files that contain no HIP call compile with g++).
The product loader stays strict; here ctypes.CDLL is wrapped so that the symbols the host-only build lacks can be declared (argtypes)
but not called."""
import ctypes
import os
import sys

import pytest

_real = ctypes.CDLL


class _Missing:
    def __call__(self, *a, **k):
        pytest.skip("symbol not in the host-only sanitizer build (it lives in the device library)")


class _Partial:
    def __init__(self, lib):
        object.__setattr__(self, "_l", lib)

    def __getattr__(self, n):
        try:
            return getattr(self._l, n)
        except AttributeError:
            return _Missing()


def _cdll(path, *a, **k):
    lib = _real(path, *a, **k)
    return _Partial(lib) if path and "elm_host_san" in str(path) else lib


ctypes.CDLL = _cdll
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
sys.exit(pytest.main(["tests/test_oracle.py", "tests/test_golden.py", "tests/test_ekf.py", "tests/test_formats.py", "tests/test_glue.py", "tests/test_correspondences.py",
                      "-q", "-x", "-rs", "-m", "not gpu", "-p", "no:cacheprovider"] + sys.argv[1:]))
