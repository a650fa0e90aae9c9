"""roctx ranges for per-layer attribution in rocprofv3 traces (tracing row): `with tracing.range("PoolLayer conv"):`
brackets the launches a layer issues; `rocprofv3 --kernel-trace --marker-trace` then shows the kernels under the layer that queued them.
A no-op unless CN_ROCTX=1 (or tracing.enable()) and the ROCm marker library is present."""
import contextlib
import ctypes
import os

_lib = None
_on = os.environ.get("CN_ROCTX", "0") != "0"


def enable(on=True):
    global _on
    _on = bool(on)


def _load():
    global _lib
    if _lib is None:
        _lib = False
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                L = ctypes.CDLL(name)
                L.roctxRangePushA.argtypes = [ctypes.c_char_p]
                L.roctxRangePushA.restype = ctypes.c_int
                L.roctxRangePop.restype = ctypes.c_int
                _lib = L
                break
            except (OSError, AttributeError):
                continue
    return _lib


@contextlib.contextmanager
def range(name, sync=None):
    """`sync`: called before the range closes WHEN TRACING IS ON (e.g. Context.sync): the range then covers the device time of the work
    it queued, not only the time to queue it - per-layer device time straight from the marker statistics of a profiling run."""
    L = _load() if _on else False
    if L:
        L.roctxRangePushA(str(name).encode())
    try:
        yield
    finally:
        if L:
            if sync is not None:
                sync()
            L.roctxRangePop()
