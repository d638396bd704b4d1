"""ctypes binding of oracle/ctc_ref.c (TEST INFRASTRUCTURE, see oracle/__init__.py).

Mirrors the call signature of the reference's Cython module
(``ctc_fast/ctc-loss/ctc_fast.pyx:13-14,154``): ``ctc_loss(params, seq, blank=0)``
with ``params`` float64 (A,T) Fortran-ordered and ``seq`` int32, returning
``(cost, grad, skip)``; ``decode_best_path(probs, blank=0) -> (hyp, align)``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsctc_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_lp = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile ctc_ref.c with gcc (a second or two)."""
    src = os.path.join(_HERE, "ctc_ref.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsctc_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.sctc_oracle_ctc_loss.restype = ctypes.c_int
        L.sctc_oracle_ctc_loss.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int,
                                           ctypes.c_int, _dp, _dp, _dp]
        L.sctc_oracle_ctc_loss_logits.restype = ctypes.c_int
        L.sctc_oracle_ctc_loss_logits.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _ip,
                                                  ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]
        L.sctc_oracle_ctc_loss_batch.restype = None
        L.sctc_oracle_ctc_loss_batch.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _ip, _lp, _ip,
                                                 _ip, _lp, ctypes.c_int, _dp, _dp, _ip,
                                                 ctypes.c_int]
        L.sctc_oracle_decode_best_path.restype = ctypes.c_int
        L.sctc_oracle_decode_best_path.argtypes = [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   _ip, _ip]
        _lib = L
    return _lib


def _check(params, seq):
    # same rejections as the Cython memoryview signature (ctc_fast.pyx:13-14)
    if not isinstance(params, np.ndarray) or params.dtype != np.float64 or params.ndim != 2:
        raise ValueError("Buffer dtype mismatch, expected 'double'")
    if not params.flags.f_contiguous:
        raise ValueError("ndarray is not Fortran contiguous")
    if not isinstance(seq, np.ndarray) or seq.dtype != np.int32 or seq.ndim != 1:
        raise ValueError("Buffer dtype mismatch, expected 'int'")
    if not seq.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")


def ctc_loss(params, seq, blank=0, return_ll_backward=False):
    _check(params, seq)
    A, T = params.shape
    grad = np.zeros((A, T), dtype=np.float64, order="F")
    cost = ctypes.c_double(0.0)
    llb = ctypes.c_double(0.0)
    rc = lib().sctc_oracle_ctc_loss(params.ctypes.data_as(_dp), A, T, seq.ctypes.data_as(_ip),
                                    seq.shape[0], int(blank), grad.ctypes.data_as(_dp),
                                    ctypes.byref(cost), ctypes.byref(llb))
    if rc < 0:
        raise ValueError("oracle ctc_loss: bad arguments (rc=%d)" % rc)
    if return_ll_backward:
        return cost.value, grad, bool(rc), llb.value
    return cost.value, grad, bool(rc)


def ctc_loss_logits(logits, seq, blank=0):
    """logits: float64 (A,T) F-order pre-softmax activations -> (cost, grad, skip, probs)."""
    _check(logits, seq)
    A, T = logits.shape
    grad = np.zeros((A, T), dtype=np.float64, order="F")
    probs = np.zeros((A, T), dtype=np.float64, order="F")
    cost = ctypes.c_double(0.0)
    rc = lib().sctc_oracle_ctc_loss_logits(logits.ctypes.data_as(_dp), A, T,
                                           seq.ctypes.data_as(_ip), seq.shape[0], int(blank),
                                           grad.ctypes.data_as(_dp), ctypes.byref(cost),
                                           probs.ctypes.data_as(_dp))
    if rc < 0:
        raise ValueError("oracle ctc_loss_logits: bad arguments (rc=%d)" % rc)
    return cost.value, grad, bool(rc), probs


def ctc_loss_batch(params_list, seq_list, blank=0, nthreads=0):
    """OpenMP-over-utterances driver (multi-core CPU baseline)."""
    B = len(params_list)
    A = params_list[0].shape[0]
    T_b = np.array([p.shape[1] for p in params_list], dtype=np.int32)
    U_b = np.array([s.shape[0] for s in seq_list], dtype=np.int32)
    foff = np.zeros(B, dtype=np.int64)
    foff[1:] = np.cumsum(T_b[:-1])
    soff = np.zeros(B, dtype=np.int64)
    soff[1:] = np.cumsum(U_b[:-1])
    y = np.concatenate([np.ascontiguousarray(p.T) for p in params_list], axis=0)  # [sumT][A]
    seq = np.concatenate(seq_list).astype(np.int32)
    grad = np.zeros_like(y)
    cost = np.zeros(B, dtype=np.float64)
    skip = np.zeros(B, dtype=np.int32)
    lib().sctc_oracle_ctc_loss_batch(y.ctypes.data_as(_dp), A, B, T_b.ctypes.data_as(_ip),
                                     foff.ctypes.data_as(_lp), seq.ctypes.data_as(_ip),
                                     U_b.ctypes.data_as(_ip), soff.ctypes.data_as(_lp),
                                     int(blank), grad.ctypes.data_as(_dp),
                                     cost.ctypes.data_as(_dp), skip.ctypes.data_as(_ip),
                                     int(nthreads))
    grads = [np.asfortranarray(grad[foff[b]:foff[b] + T_b[b]].T) for b in range(B)]
    return cost, grads, skip.astype(bool)


def decode_best_path(probs, blank=0):
    if not isinstance(probs, np.ndarray) or probs.dtype != np.float64 or probs.ndim != 2:
        raise ValueError("Buffer dtype mismatch, expected 'double'")
    if not probs.flags.f_contiguous:
        raise ValueError("ndarray is not Fortran contiguous")
    A, T = probs.shape
    hyp = np.zeros(T, dtype=np.int32)
    align = np.zeros(T, dtype=np.int32)
    n = lib().sctc_oracle_decode_best_path(probs.ctypes.data_as(_dp), A, T, int(blank),
                                           hyp.ctypes.data_as(_ip), align.ctypes.data_as(_ip))
    if n < 0:
        raise IndexError("list assignment index out of range")
    return [int(v) for v in hyp[:n]], [int(v) for v in align[:n]]
