"""pdwt_amd -- MI355X-native (gfx950) implementation of PDWT's separable DWT hot path.

The product is native: hand-written HIP kernels behind a C-ABI (include/pdwt_hip.h,
pdwt_amd/lib/libpdwt_hip.so) and the reference's own C++ ``Wavelets`` class above it
(include/wt.h, libpdwt.so / libpdwtd.so).  This Python package is a thin ctypes view of those
libraries for tests, bench.py and Python callers -- the shape of the reference's external pypwt
binding (README.md:24).
"""
from ._native import Info, hip, host, require_gpu  # noqa: F401
from .wavelets import DeviceArray, ImageBatch, Wavelets, W_CREATION_ERROR, W_FORWARD, W_INIT, W_INVERSE  # noqa: F401

__all__ = ["Wavelets", "ImageBatch", "DeviceArray", "Info", "hip", "host", "require_gpu"]
