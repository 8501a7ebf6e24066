"""edge_alignment_amd — MI355X-native hot path of direct edge alignment (kuwt/edge_alignment).

The product is the C-ABI shared library `lib/libea_hip.so` (include/ea_hip.h) and the C++ shim
headers in `include/`; this Python package only holds the build driver, the ctypes stub and
the synthetic-workload generators used by tests/ and bench.py.
"""
from . import capi  # noqa: F401
from .build import build_library  # noqa: F401

__all__ = ["capi", "build_library"]
