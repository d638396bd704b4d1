#!/usr/bin/env python3
"""scratch GPU debugging (not a test)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa
import ctc_fast as cf  # noqa
from oracle import ctc as octc  # noqa
from oracle import brnn as obrnn  # noqa
from tests.helpers import softmax0  # noqa

np.seterr(all="ignore")


def ctc_cases():
    rs = np.random.RandomState(11)
    A = 33
    shapes = [(300, 30), (257, 40), (64, 31), (1000, 100), (5, 2), (129, 64), (1, 1)]
    probs, seqs = [], []
    for T, U in shapes:
        probs.append(np.asfortranarray(softmax0(rs.randn(A, T) * 2.0)))
        s = rs.randint(0, A, size=U).astype(np.int32)
        if U > 3:
            s[1] = s[2]
        seqs.append(s)
    refs = [octc.ctc_loss(np.asfortranarray(p.astype(np.float32).astype(np.float64)), s) for p, s in zip(probs, seqs)]
    for dt in (np.float32, np.float64):
        costs, grads, skips = cf.ctc_loss_batch([p.astype(dt) for p in probs], seqs)
        print("batched", dt.__name__, ["%.4f/%.4f" % (c, r[0]) for c, r in zip(costs, refs)])
        for i, (p, s) in enumerate(zip(probs, seqs)):
            c1, g1, k1 = cf.ctc_loss_batch([p.astype(dt)], [s])
            print("  single %d T=%d U=%d: %.6f ref %.6f skip %s/%s graderr %.2e" %
                  (i, p.shape[1], len(s), c1[0], refs[i][0], k1[0], refs[i][2],
                   np.abs(g1[0] - refs[i][1]).max()))
    # same utterance 0 with labels in 1..A-1 and no repeat, f32
    p, s = probs[0], seqs[0].copy()
    print("seq0", s)
    s2 = np.where(s == 0, 5, s).astype(np.int32)
    c, _, _ = cf.ctc_loss_batch([p.astype(np.float32)], [s2])
    print("  no-blank-label f32: %.6f ref %.6f" % (c[0], octc.ctc_loss(p, s2)[0]))
    # pad the batch with a long-U utterance to force K=4
    c, _, _ = cf.ctc_loss_batch([p.astype(np.float32), probs[3].astype(np.float32)], [s2, seqs[3]])
    print("  no-blank-label f32 with K=4: %.6f" % c[0])
    c, _, _ = cf.ctc_loss_batch([p.astype(np.float64), probs[3].astype(np.float64)], [s2, seqs[3]])
    print("  no-blank-label f64 with K=4: %.6f" % c[0])


def rec_cases():
    from nnets import brnnet
    D, A, H, NL, TL = 64, 33, 1824, 2, 1
    B, T = 32, 60
    rs = np.random.RandomState(3)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    datas = [rs.randn(D, T) for _ in range(B)]
    labs = [rs.randint(1, A, size=6).astype(np.int32) for _ in range(B)]
    st = [[w, b] for w, b in zip(params["W"], params["b"])] + [[params["Wf"], None], [params["Wb"], None]]
    ref_c, ref_g, _, _ = obrnn.cost_and_grad_batch(params, datas[:4], labs[:4], TL)
    for sync in (0, 1):
        os.environ["SCTC_REC_SYNC"] = str(sync)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
        net.setParams(st)
        outs = []
        for rep in range(3):
            c, g, s = net.costAndGradBatch(datas, labs)
            outs.append((c.copy(), net.grad[NL + 1][0].copy_to_host().copy()))
        print("sync", sync, "cost[:4]", outs[0][0][:4], "ref", ref_c)
        print("   rep diffs cost", np.abs(outs[0][0] - outs[1][0]).max(), np.abs(outs[0][0] - outs[2][0]).max(),
              "dWf", np.abs(outs[0][1] - outs[1][1]).max())
        c4, g4, s4 = net.costAndGradBatch(datas[:4], labs[:4])
        gw = net.grad[NL + 1][0].copy_to_host()
        print("   B=4 cost", c4, "dWf relerr", np.linalg.norm(gw - ref_g["Wf"]) / np.linalg.norm(ref_g["Wf"]))
        del net


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["ctc", "rec"]):
        {"ctc": ctc_cases, "rec": rec_cases}[name]()
