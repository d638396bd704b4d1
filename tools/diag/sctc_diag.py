"""ctypes loader of tools/diag/libsctc_diag.so (hardware probes; not the product library)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsctc_diag.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            import torch  # noqa: F401  (same HIP runtime instance as the product library)
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        f32p = ctypes.POINTER(ctypes.c_float)
        L.sctc_diag_last_error.restype = ctypes.c_char_p
        L.sctc_selftest.argtypes = [ctypes.c_void_p]
        L.sctc_probe_fabric.argtypes = [f32p, ctypes.c_int32, ctypes.c_void_p]
        L.sctc_probe_mfma.argtypes = [f32p, ctypes.c_int32, ctypes.c_void_p]
        L.sctc_diag_spin.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
        L.sctc_probe_handoff.argtypes = [f32p, ctypes.c_int32, ctypes.c_void_p]
        L.sctc_diag_stream_cu_mask.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int32,
                                               ctypes.POINTER(ctypes.c_void_p)]
        L.sctc_diag_stream_destroy.argtypes = [ctypes.c_void_p]
        L.sctc_diag_where.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_void_p]
        _lib = L
    return _lib
