#!/usr/bin/env python3
"""The long-row CTC kernel (ctc_fusedw.hip, rows of 513..2048 states) against the oracle and the lattice + grad kernels
of rounds 1-5 on one shape; prints errors and, with --time, milliseconds per call (events around the C entry).
usage: tools/ctc_wide_probe.py [--shape A,T,U] [--B n] [--dtype f32|f64] [--time]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
import ctc_fast  # noqa: E402
from oracle import ctc as octc  # noqa: E402
from tests.helpers import softmax0  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="33,8000,800")
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--time", action="store_true")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
A, T, U = [int(v) for v in args.shape.split(",")]
dt = np.float32 if args.dtype == "f32" else np.float64
rs = np.random.RandomState(1)
probs, seqs = [], []
for b in range(args.B):
    probs.append(np.asfortranarray(softmax0(rs.randn(A, T)).astype(dt)))
    seqs.append(rs.randint(1, A, size=U).astype(np.int32))
with np.errstate(all="ignore"):
    ref = octc.ctc_loss(np.asfortranarray(probs[0].astype(np.float64)), seqs[0])
res = {}
for name, env in (("wide", {"SCTC_CTC_WIDE_MIN_B": "1"}), ("lattice", {"SCTC_CTC_WIDE": "0"})):
    for k in ("SCTC_CTC_WIDE", "SCTC_CTC_WIDE_MIN_B"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with np.errstate(all="ignore"):
        cost, grads, skip = ctc_fast.ctc_loss_batch(probs, seqs)
    res[name] = (cost, grads, skip)
    err = np.abs(grads[0].astype(np.float64) - ref[1]).max()
    print("%-8s cost %.9f (oracle %.9f, rel %.1e) skip %s (oracle %s) |grad - oracle| %.2e  costs %s" % (
        name, cost[0], ref[0], abs(cost[0] - ref[0]) / max(abs(ref[0]), 1e-300), list(skip), ref[2], err,
        np.array2string(cost[:4], precision=4)), flush=True)
    if args.time:
        dev = torch.from_numpy(np.concatenate([np.ascontiguousarray(p.T) for p in probs], axis=0)).cuda()
        lengths = [T] * args.B
        for _ in range(2):
            ctc_fast.ctc_loss_batch(dev, seqs, lengths=lengths)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctc_fast.ctc_loss_batch(dev, seqs, lengths=lengths)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("%-8s %.3f ms per call (best of %d, events around ctc_loss_batch incl. its uploads)" % (name, best, args.reps), flush=True)
print("wide vs lattice: max |dgrad| %.2e" % max(np.abs(a.astype(np.float64) - b).max() for a, b in zip(res["wide"][1], res["lattice"][1])))
