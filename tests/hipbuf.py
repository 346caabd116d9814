"""Device buffers for the GPU tests through the HIP runtime directly (ctypes): the parity tests then need neither
torch nor its minute-long first import on a fresh box.  Test infrastructure only."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


def synchronize():
    _check(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")


class DeviceBuffer:
    """`nbytes` of device memory; `.ptr` is the raw device address."""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        _check(hip().hipMalloc(C.byref(p), max(int(nbytes), 1)), "hipMalloc")
        self.ptr, self.nbytes = p.value, int(nbytes)

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        _check(hip().hipMemcpy(b.ptr, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        return b

    def to_numpy(self, dtype=np.uint8) -> np.ndarray:
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(hip().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
