"""Shared test helpers: synthetic generators and brute-force CTC."""
import itertools

import numpy as np


def softmax0(x):
    e = np.exp(x - x.max(axis=0, keepdims=True))
    return e / e.sum(axis=0, keepdims=True)


def time_trials_input():
    """ctc/time_trials.py:13-25 (seed 33, A=40, U=125, T=1200, peaked)."""
    np.random.seed(33)
    A, U, T = 40, 125, 1200
    seq = np.floor(np.random.rand(U) * A).astype(np.int32)
    p = np.random.randn(A, T)
    p[seq, np.arange(U)] = 3
    p[0, U:] = 3
    p = np.exp(p)
    p = p / np.sum(p, axis=0)
    return p, seq


def mid_input(T, A, U, seed):
    rs = np.random.RandomState(seed)
    logits = rs.randn(A, T)
    seq = rs.randint(1, A, size=U).astype(np.int32)
    return logits, seq


def collapse(path, blank=0):
    out = []
    prev = None
    for k in path:
        if k != prev and k != blank:
            out.append(k)
        prev = k
    return out


def brute_force_ctc(y, seq, blank=0):
    """-ln sum over all A^T frame labellings that collapse to seq (tiny cases only).
    Only meaningful when no label equals the blank id."""
    A, T = y.shape
    total = 0.0
    target = list(seq)
    for path in itertools.product(range(A), repeat=T):
        if collapse(path, blank) == target:
            p = 1.0
            for t, k in enumerate(path):
                p *= y[k, t]
            total += p
    return -np.log(total)


def fd_grad_logits(fn, logits, seq, eps=1e-5):
    """central finite difference of cost wrt logits; fn(logits, seq) -> cost"""
    g = np.zeros_like(logits)
    for k in range(logits.shape[0]):
        for t in range(logits.shape[1]):
            lp = logits.copy()
            lm = logits.copy()
            lp[k, t] += eps
            lm[k, t] -= eps
            g[k, t] = (fn(lp, seq) - fn(lm, seq)) / (2 * eps)
    return g


def load_net(npz, prefix=""):
    """fixture -> (params dict for oracle.brnn, grads dict, dims, data, labels, cost)"""
    n = int(npz[prefix + "n"])
    D, A, H, NL, TL, T = [int(v) for v in npz[prefix + "dims"]]
    W = [npz[prefix + "W%d" % i] for i in range(NL + 1)]
    b = [npz[prefix + "b%d" % i] for i in range(NL + 1)]
    dW = [npz[prefix + "dW%d" % i] for i in range(NL + 1)]
    db = [npz[prefix + "db%d" % i] for i in range(NL + 1)]
    params = {"W": W, "b": b, "Wf": None, "Wb": None}
    grads = {"W": dW, "b": db, "Wf": None, "Wb": None}
    if n == NL + 3:
        params["Wf"] = npz[prefix + "W%d" % (NL + 1)]
        params["Wb"] = npz[prefix + "W%d" % (NL + 2)]
        grads["Wf"] = npz[prefix + "dW%d" % (NL + 1)]
        grads["Wb"] = npz[prefix + "dW%d" % (NL + 2)]
    return (params, grads, (D, A, H, NL, TL, T), npz[prefix + "data"], npz[prefix + "labels"],
            float(npz[prefix + "cost"]))


_ORACLE_CTX = None     # (params, datas, labs, TL, max_act, want_grad): inherited by the forked workers


def _oracle_chunk(idx):
    """worker of oracle_parallel: the utterances `idx` of the inherited problem; returns their
    costs / skips and the SUM of their gradients (one pickled gradient per worker, not per
    utterance)"""
    params, datas, labs, TL, max_act, want_grad, mixed_rec, masks_of = _ORACLE_CTX
    from oracle import brnn as obrnn
    mixed = None if mixed_rec is None else obrnn.Mixed(rec=mixed_rec)
    from oracle import ctc as octc
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=job_threads())
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
    costs, skips, total = [], [], None
    with ctx, np.errstate(all="ignore"):
        for i in idx:
            data = np.asarray(datas[i], dtype=np.float64)
            if want_grad:
                c, g, s, _ = obrnn.cost_and_grad(params, data, labs[i], TL, max_act, mixed=mixed,
                                                 masks=masks_of(i) if masks_of else None)
            else:
                logits, _ = obrnn.forward(params, data, TL, max_act, mixed)
                c, _, s = octc.ctc_loss(np.asfortranarray(obrnn.softmax_cols(logits)),
                                        np.ascontiguousarray(labs[i], dtype=np.int32), 0)
                g = None
            costs.append(c)
            skips.append(bool(s))
            if g is None or s:
                continue
            if total is None:
                total = g
            else:
                for a, b in zip(total["W"], g["W"]):
                    a += b
                for a, b in zip(total["b"], g["b"]):
                    a += b
                total["Wf"] += g["Wf"]
                total["Wb"] += g["Wb"]
    return list(idx), costs, skips, total


def job_threads():
    import os
    return max(1, int(os.environ.get("SCTC_ORACLE_THREADS", "8")))


def oracle_parallel(params, datas, labs, TL, max_act=20.0, want_grad=True, procs=None,
                    mixed_rec=None, masks_of=None):
    """the float64 oracle over a list of utterances in forked worker processes (a few BLAS
    threads each): the full-size configurations take seconds instead of minutes on the GPU
    box's host.  Returns (costs, summed gradient dict or None, skips).  mixed_rec: None = exact
    float64; True / False = oracle.brnn.Mixed(rec=...) (the 16-bit-operand numerics).  masks_of:
    callable i -> the ReLU / (0,maxAct) masks utterance i's backward pass must use (the device's own
    decisions, oracle.brnn.cost_and_grad `masks`); evaluated inside the workers."""
    global _ORACLE_CTX
    import multiprocessing as mp
    import os
    n = len(datas)
    procs = procs or max(1, min(n, 16, (os.cpu_count() or 8) // job_threads()))
    _ORACLE_CTX = (params, datas, labs, TL, max_act, want_grad, mixed_rec, masks_of)
    chunks = [list(range(k, n, procs)) for k in range(procs)]
    try:
        if procs == 1:
            res = [_oracle_chunk(chunks[0])]
        else:
            with mp.get_context("fork").Pool(procs) as pool:
                res = pool.map(_oracle_chunk, chunks, chunksize=1)
    finally:
        _ORACLE_CTX = None
    costs = np.zeros(n)
    skips = np.zeros(n, dtype=bool)
    total = None
    for idx, c, s, g in res:
        costs[idx] = c
        skips[idx] = s
        if g is None:
            continue
        if total is None:
            total = g
        else:
            for a, b in zip(total["W"], g["W"]):
                a += b
            for a, b in zip(total["b"], g["b"]):
                a += b
            total["Wf"] += g["Wf"]
            total["Wb"] += g["Wb"]
    return costs, total, skips


_VARIANT_CTX = None


def _oracle_variant(k):
    params, data, lab, TL, max_act, jobs = _VARIANT_CTX
    from oracle import brnn as obrnn
    job = jobs[k]
    mixed = None if job.get("mixed_rec") is None else obrnn.Mixed(rec=job["mixed_rec"])
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=job_threads())
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx, np.errstate(all="ignore"):
        c, g, s, _ = obrnn.cost_and_grad(params, np.asarray(data, dtype=np.float64), lab, TL, max_act,
                                         mixed=mixed, masks=job.get("masks"))
    return c, g, bool(s)


def oracle_variants(params, data, lab, TL, jobs, max_act=20.0):
    """ONE utterance through several variants of the float64 oracle at once, one forked process each:
    jobs = [{"mixed_rec": None | False | True, "masks": None | device masks}, ...]
    -> [(cost, grads, skip), ...].  The cfg-5 utterance (T = 8000, 7 x 2048) takes the host about a
    minute per variant; plain + device-gated (or Mixed + exact) run side by side."""
    global _VARIANT_CTX
    import multiprocessing as mp
    _VARIANT_CTX = (params, data, lab, TL, max_act, jobs)
    try:
        if len(jobs) == 1:
            return [_oracle_variant(0)]
        with mp.get_context("fork").Pool(len(jobs)) as pool:
            return pool.map(_oracle_variant, range(len(jobs)), chunksize=1)
    finally:
        _VARIANT_CTX = None
