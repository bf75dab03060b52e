"""Runs the CPU test files against the AddressSanitizer + UndefinedBehaviorSanitizer builds of the CPU code of this repository (make -C oracle san):
the checker (oracle/gs4d_oracle.cpp) and the product's host library (host/gs4d_host.cpp — plain C++, no HIP).  Started by tests/test_sanitizers.py
in a python of its own with the sanitizer runtimes preloaded; the HIP build is never instrumented (GPU sanitizers are not available on this pool).

The product binding (4dgaussiansplatrendering_amd/__init__.py) is imported as it is; its gs4d_host_* entry points are then redirected to the sanitized
host library, with the argument types the binding declared."""
import ctypes
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SAN = os.path.join(ROOT, "oracle", "_san")
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ["GS4D_ORACLE_SO"] = os.path.join(SAN, "libgs4d_oracle_san.so")

import pytest  # noqa: E402


class HostProxy:
    def __init__(self, real, san):
        self._real, self._san = real, san

    def __getattr__(self, name):
        real = getattr(self._real, name)
        if not name.startswith("gs4d_host_"):
            return real
        f = getattr(self._san, name)
        f.restype, f.argtypes = real.restype, real.argtypes
        return f


def main():
    pkg = importlib.import_module("4dgaussiansplatrendering_amd")
    pkg._lib = HostProxy(pkg._lib, ctypes.CDLL(os.path.join(SAN, "libgs4d_host_san.so")))
    files = sys.argv[1:] or ["test_oracle_golden.py", "test_oracle_sort.py", "test_oracle_render.py", "test_oracle_gl.py", "test_oracle_splat_draw.py", "test_host_math.py"]
    return pytest.main(["-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"] + [os.path.join(HERE, f) for f in files])


if __name__ == "__main__":
    sys.exit(main())
