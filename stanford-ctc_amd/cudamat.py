"""The slice of the `cudamat` surface that the reference's trainer touches on weight,
gradient and velocity objects (sgd.py:21-23,39-41,52-55,105-106,136-139;
runNNet.py:117-120; brnnet.py:13,255-256,264-276) -- an API facade (names only) over
HIP buffers and libsctc_hip.so.  No cudamat / CUDA code is involved.

A CUDAMatrix is a device matrix of logical shape (rows, cols) stored row-major and
zero-padded to [ceil32(rows)][ceil64(cols)] (a (rows,1) vector: [ceil32(rows)]), the
layout of the BRNN engine's flat parameter buffer, so same-shape matrices can be
combined with one contiguous kernel.  PyTorch only provides the allocation.
"""
import ctypes

import numpy as np

import _sctc


def _pad32(n):
    return (int(n) + 31) // 32 * 32


def padded_layout(rows, cols):
    """-> (rows_p, ld) of a logical (rows, cols) matrix: rows padded to 32, the row stride to
    64 floats (256 B, see LD() in csrc/brnn_engine.hip)"""
    if cols == 1:
        return _pad32(rows), 1
    return _pad32(rows), (int(cols) + 63) // 64 * 64


def cuda_set_device(n):
    torch = _sctc.require_gpu()
    torch.cuda.set_device(int(n))
    _sctc.check(_sctc.lib().sctc_set_device(int(n)), "cuda_set_device")


def cublas_init():
    _sctc.lib()      # nothing to initialise; fail here if the library is missing


def shutdown():
    pass


class CUDAMatrix(object):
    def __init__(self, array=None, _flat=None, _shape=None):
        torch = _sctc.require_gpu()
        if _flat is not None:
            self.shape = tuple(_shape)
            self._flat = _flat
            self.numpy_array = None
        else:
            array = np.asarray(array)
            if array.ndim == 1:
                array = array.reshape(-1, 1)
            self.shape = tuple(array.shape)
            rows_p, ld = padded_layout(*self.shape)
            self._flat = torch.zeros(rows_p * ld, dtype=torch.float32, device="cuda")
            self.numpy_array = np.array(array, dtype=np.float32)
            self.copy_to_device()
        self._ws = None

    # -- layout helpers
    def _view2d(self):
        rows_p, ld = padded_layout(*self.shape)
        return self._flat.view(rows_p, ld)[:self.shape[0], :self.shape[1]]

    # -- host <-> device (sgd.py:39-41,52-55; brnnet.py:264-276)
    def copy_to_host(self):
        self.numpy_array = self._view2d().cpu().numpy().copy()
        return self.numpy_array

    def copy_to_device(self):
        torch = _sctc.require_gpu()
        self._view2d().copy_(torch.from_numpy(np.ascontiguousarray(self.numpy_array,
                                                                   dtype=np.float32)))

    def asarray(self):
        return self.copy_to_host()

    def assign(self, value):
        if isinstance(value, CUDAMatrix):
            self._flat.copy_(value._flat)
        else:
            self._flat.fill_(0.0)
            self._view2d().fill_(float(value))
        return self

    # -- arithmetic used by sgd.py / NNet.updateParams
    def mult(self, alpha):
        _sctc.check(_sctc.lib().sctc_scale(self._flat.data_ptr(), float(alpha),
                                           self._flat.numel(), _sctc.current_stream_ptr()),
                    "mult")
        return self

    def add_mult(self, other, alpha=1.0):
        if other.shape != self.shape:
            raise ValueError("add_mult: shape mismatch %s vs %s" % (self.shape, other.shape))
        _sctc.check(_sctc.lib().sctc_axpy(self._flat.data_ptr(), other._flat.data_ptr(),
                                          float(alpha), self._flat.numel(),
                                          _sctc.current_stream_ptr()), "add_mult")
        return self

    def euclid_norm(self):
        torch = _sctc.require_gpu()
        if self._ws is None:
            self._ws = torch.empty(8192 + 8, dtype=torch.uint8, device="cuda")
        out = self._ws[8192:].view(torch.float64)
        _sctc.check(_sctc.lib().sctc_sumsq(self._flat.data_ptr(), self._flat.numel(),
                                           out.data_ptr(), self._ws.data_ptr(), 8192,
                                           _sctc.current_stream_ptr()), "euclid_norm")
        return float(np.sqrt(out.item()))


def empty(shape):
    torch = _sctc.require_gpu()
    rows_p, ld = padded_layout(*shape)
    return CUDAMatrix(_flat=torch.zeros(rows_p * ld, dtype=torch.float32, device="cuda"),
                      _shape=shape)
