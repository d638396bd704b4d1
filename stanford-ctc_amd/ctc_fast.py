"""Drop-in for the reference's Cython extension module ``ctc_fast``
(``ctc_fast/ctc-loss/ctc_fast.pyx``, built by ``ctc_fast/ctc-loss/setup.py:7-8``),
running the alpha/beta recursion and the gradient on the MI355X through
libsctc_hip.so.  Same surface, same argument checking, same return values:

    ctc_loss(params, seq, blank=0) -> (cost, grad, skip)      ctc_fast.pyx:13-152
    decode_best_path(probs, blank=0) -> (hyp, align)          ctc_fast.pyx:154-187

``params``/``probs`` are float64 (A, T) Fortran-ordered NumPy arrays and ``seq`` a
C-contiguous int32 vector, exactly what the Cython memoryview signature accepts;
anything else raises ``ValueError`` like the memoryview does.  The float64 host
signature runs the float64 device kernels.  Additional entry points (not in the
reference) take batches and device tensors: :func:`ctc_loss_batch`.

There is no CPU fallback: without the HIP library or without a GPU the calls raise.
"""
import ctypes

import numpy as np

import _sctc

# ctc_fast.pyx:6 -- the reference flips NumPy's error state process-wide at import
np.seterr(divide='raise', invalid='raise')


def _check_params(params, name="params"):
    if params is None:
        raise TypeError("Argument '%s' must not be None" % name)
    if not isinstance(params, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray)" % name)
    if params.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % params.ndim)
    if params.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'double' but got '%s'" % params.dtype)
    if not params.flags.f_contiguous:
        raise ValueError("ndarray is not Fortran contiguous")


def _check_seq(seq):
    if seq is None:
        raise TypeError("Argument 'seq' must not be None")
    if not isinstance(seq, np.ndarray):
        raise TypeError("Argument 'seq' has incorrect type (expected numpy.ndarray)")
    if seq.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % seq.ndim)
    if seq.dtype != np.int32:
        raise ValueError("Buffer dtype mismatch, expected 'int' but got '%s'" % seq.dtype)
    if not seq.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")


def _run_batch(probs_dev, grad_dev, A, ld, blank, T_b, U_b, frame_off, labels, label_off,
               dtype, rowbase_dev=None):
    """probs_dev/grad_dev: torch CUDA tensors [rows][ld]; returns (cost, skip) torch tensors."""
    torch = _sctc.require_gpu()
    L = _sctc.lib()
    B = len(T_b)
    T_b = np.ascontiguousarray(T_b, dtype=np.int32)
    U_b = np.ascontiguousarray(U_b, dtype=np.int32)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    label_off = np.ascontiguousarray(label_off, dtype=np.int64)
    bt = _sctc.CtcBatch(B, int(A), int(blank), dtype, int(ld), _sctc.i32(T_b), _sctc.i32(U_b),
                        _sctc.i64(frame_off), _sctc.i32(labels), _sctc.i64(label_off),
                        ctypes.c_void_p(rowbase_dev.data_ptr() if rowbase_dev is not None else 0))
    nbytes = L.sctc_ctc_workspace_bytes(ctypes.byref(bt))
    if nbytes == 0:
        _sctc.check(-1, "ctc_loss")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=probs_dev.device)
    cost = torch.empty(B, dtype=torch.float64, device=probs_dev.device)
    skip = torch.empty(B, dtype=torch.int32, device=probs_dev.device)
    rc = L.sctc_ctc_loss_batch(ctypes.byref(bt), probs_dev.data_ptr(), grad_dev.data_ptr(),
                               cost.data_ptr(), skip.data_ptr(), ws.data_ptr(), nbytes,
                               _sctc.current_stream_ptr())
    _sctc.check(rc, "ctc_loss")
    return cost, skip


def ctc_loss(params, seq, blank=0):
    """CTC loss function (ctc_fast.pyx:13-152).

    params - n x m matrix of n-D probability distributions over m frames, float64,
    Fortran order.  seq - int32 label ids.  Returns (cost, grad, skip): the negative
    log-likelihood, its gradient with respect to the *unnormalised* (pre-softmax)
    activations as a fresh float64 (n, m) Fortran array, and the skip flag the
    reference sets when a frame normaliser is zero (ctc_fast.pyx:147-149).
    """
    _check_params(params)
    _check_seq(seq)
    torch = _sctc.require_gpu()
    A, T = params.shape
    if seq.shape[0] == 0:
        # undefined behaviour in the reference (reads seq[0] out of bounds, ctc_fast.pyx:43)
        raise ValueError("ctc_loss: empty label sequence")
    blank = int(blank)
    if blank < 0:
        raise OverflowError("can't convert negative value to unsigned int")
    dev_probs = torch.from_numpy(params.T).cuda()          # (T, A) row-major == (A,T) F-order
    dev_grad = torch.empty_like(dev_probs)
    cost, skip = _run_batch(dev_probs, dev_grad, A, A, blank, [T], [seq.shape[0]], [0], seq, [0],
                            _sctc.F64)
    grad = np.asfortranarray(dev_grad.cpu().numpy().T)
    return float(cost.item()), grad, bool(skip.item())


def ctc_loss_batch(probs, seqs, blank=0, lengths=None):
    """Batched form (not in the reference).

    probs: list of (A, T_b) float64/float32 F-ordered arrays, or a torch CUDA tensor
    [sum T][A] (float32/float64) with ``lengths`` giving T_b.  Returns
    (cost float64[B], grad in the input's form, skip bool[B]).
    """
    torch = _sctc.require_gpu()
    B = len(seqs)
    U_b = [len(s) for s in seqs]
    labels = np.concatenate([np.asarray(s, dtype=np.int32) for s in seqs])
    label_off = np.concatenate([[0], np.cumsum(U_b)[:-1]])
    as_list = not isinstance(probs, torch.Tensor)
    if as_list:
        dt = probs[0].dtype
        T_b = [p.shape[1] for p in probs]
        A = probs[0].shape[0]
        host = np.concatenate([np.ascontiguousarray(np.asarray(p).T) for p in probs], axis=0)
        dev = torch.from_numpy(host).cuda()
    else:
        dev = probs.contiguous()
        T_b = list(lengths)
        A = dev.shape[1]
        dt = np.float64 if dev.dtype == torch.float64 else np.float32
    if dev.dtype not in (torch.float32, torch.float64):
        raise ValueError("Buffer dtype mismatch, expected 'double' or 'float'")
    frame_off = np.concatenate([[0], np.cumsum(T_b)[:-1]])
    grad = torch.empty_like(dev)
    cost, skip = _run_batch(dev, grad, A, dev.shape[1], blank, T_b, U_b, frame_off, labels,
                            label_off, _sctc.F64 if dev.dtype == torch.float64 else _sctc.F32)
    if as_list:
        g = grad.cpu().numpy()
        grads = [np.asfortranarray(g[o:o + t].T.astype(dt)) for o, t in zip(frame_off, T_b)]
        return cost.cpu().numpy(), grads, skip.cpu().numpy().astype(bool)
    return cost, grad, skip.bool()


def decode_best_path(probs, blank=0):
    """Best path decoding (ctc_fast.pyx:154-187): most likely label per frame
    (argmax on the GPU), then drop blanks, drop the reference's hard-coded ids
    1, 2 and 8 (ctc_fast.pyx:176-179), collapse repeats.  Returns (hyp, align)."""
    _check_params(probs, "probs")
    torch = _sctc.require_gpu()
    A, T = probs.shape
    dev = torch.from_numpy(probs.T).cuda()
    best = torch.empty(T, dtype=torch.int32, device=dev.device)
    rc = _sctc.lib().sctc_argmax_rows(dev.data_ptr(), _sctc.F64, best.data_ptr(), T, A, A,
                                      _sctc.current_stream_ptr())
    _sctc.check(rc, "decode_best_path")
    return collapse_best_path(best.cpu().numpy(), blank)


def collapse_best_path(best_path, blank=0):
    hyp, align = [], []
    for i in range(len(best_path)):
        b = int(best_path[i])
        if b == blank:
            continue
        if b == 1 or b == 2 or b == 8:
            continue
        elif i != 0 and b == best_path[i - 1]:
            align[-1] = i
            continue
        else:
            hyp.append(b)
            align.append(i)
    return hyp, align
