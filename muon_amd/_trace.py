"""roctx ranges around the phases of the hot path (tracing / profiling).

``MUON_AMD_TRACE=1`` loads ``libroctx64.so`` (ROCm's marker library) and brackets the phases of ``tfidf_device`` and
``lsi_device`` - and the iterations of the MOFA engines - with ``roctxRangePushA`` / ``roctxRangePop``, so that a
``rocprofv3 --marker-trace --kernel-trace`` run shows which kernels belong to which phase.  Off (the default) it costs
one attribute lookup per phase; a missing library turns it off with a warning instead of failing the call.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import warnings

_lib = None
_state = None  # None: not decided yet; False: off; True: on


def _enabled() -> bool:
    global _lib, _state
    if _state is None:
        _state = False
        if os.environ.get("MUON_AMD_TRACE", "0") == "1":
            for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
                try:
                    _lib = ctypes.CDLL(name)
                    _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    _lib.roctxRangePushA.restype = ctypes.c_int
                    _lib.roctxRangePop.restype = ctypes.c_int
                    _state = True
                    break
                except (OSError, AttributeError):
                    _lib = None
            if not _state:
                warnings.warn("MUON_AMD_TRACE=1 but libroctx64.so could not be loaded: no ranges are emitted")
    return bool(_state)


@contextlib.contextmanager
def phase(name: str):
    """``with phase("lsi/expand"):`` - a roctx range when tracing is on, nothing otherwise."""
    if not _enabled():
        yield
        return
    _lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        _lib.roctxRangePop()


def mark(name: str) -> None:
    """Close the innermost open range of this module's stack and open ``name`` (phases that follow each other)."""
    if _enabled():
        if _depth[0] > 0:
            _lib.roctxRangePop()
            _depth[0] -= 1
        if name:
            _lib.roctxRangePushA(name.encode())
            _depth[0] += 1


_depth = [0]
