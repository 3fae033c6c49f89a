"""Tiny ctypes view of the HIP runtime already loaded by PyTorch-ROCm (debug copies only)."""
import ctypes

_rt = None


def _runtime():
    global _rt
    if _rt is None:
        import torch  # noqa: F401  (loads its bundled libamdhip64, SONAME libamdhip64.so.7)
        _rt = ctypes.CDLL("libamdhip64.so.7")
        _rt.hipMemcpy.restype = ctypes.c_int
        _rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return _rt


def memcpy_d2h(host_buf, dev_ptr, nbytes):
    host = host_buf if isinstance(host_buf, ctypes.c_void_p) else ctypes.cast(host_buf, ctypes.c_void_p)
    rc = _runtime().hipMemcpy(host, ctypes.c_void_p(dev_ptr), nbytes, 2)
    if rc != 0:
        raise RuntimeError("hipMemcpy D2H failed: %d" % rc)
