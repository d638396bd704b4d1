"""A/B of recurrent-kernel variants on the headline shapes (cfg-3, minibatch 32): the variants named
on the command line (SCTC_REC_VARIANT values) must give bit-identical costs and gradients; prints
step time and phase times of each.  usage: gpu_ab_rec.py [T=1000] [variants=0,4] [B=32]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _sctc  # noqa: E402
from nnets import brnnet  # noqa: E402

PHASES = ["fwd_gemm", "fwd_rec", "ctc", "bwd_gemm", "bwd_rec", "other"]


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    variants = [v for v in (sys.argv[2] if len(sys.argv) > 2 else "0,4").split(",")]
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    D, A, H, NL, TL, U = 483, 33, 1824, 5, 3, max(1, T // 10)
    rs = np.random.RandomState(1)
    feats = torch.randn(B * T, D, device="cuda")
    labels = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    Ts = [T] * B
    L = _sctc.lib()
    ref = None
    for v in variants:
        os.environ["SCTC_REC_VARIANT"] = v
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
        net.initParams()
        cost, _, skip = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
        g = net.grad.flat.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        L.sctc_brnn_set_profiling(net._h, 1)
        acc = np.zeros(6)
        arr = (ctypes.c_float * 6)()
        for _ in range(3):
            net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
            L.sctc_brnn_phase_ms(net._h, arr)
            acc += np.array(list(arr))
        acc /= 3
        L.sctc_brnn_set_profiling(net._h, 0)
        same = "-" if ref is None else ("bit-identical" if (torch.equal(g, ref[0]) and np.array_equal(cost, ref[1])) else
                                        "DIFFERENT: max |dg| %.3e rel cost %.3e" % (float((g - ref[0]).abs().max()), float(np.abs(cost - ref[1]).max() / np.abs(ref[1]).max())))
        if ref is None:
            ref = (g, cost)
        print("variant %s: path %s  step %.2f ms  %.0f frames/s  rec us/step fwd %.2f bptt %.2f | %s | vs first: %s"
              % (v, net.recurrentPath(), ms, B * T / ms * 1e3, acc[1] * 1e3 / (T - 1), acc[4] * 1e3 / (T - 1),
                 ", ".join("%s %.2f" % (k, x) for k, x in zip(PHASES, acc)), same))
        del net
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
