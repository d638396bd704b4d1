"""Timing / agreement of the CTC lattice shapes (waves x states per lane) on long label rows:
prints ms per call, the costs and a gradient checksum.  usage: gpu_ctc_shape.py  (run once per library /
SCTC_CTC_WAVES setting and compare the lines)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import ctc_fast  # noqa: E402


def main():
    rs = np.random.RandomState(0)
    for (B, T, U, A) in ((1, 8000, 800, 33), (8, 8000, 800, 33), (4, 4000, 600, 33), (4, 4000, 400, 33), (32, 3000, 300, 33)):
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        probs = torch.softmax(torch.randn(B * T, A, device="cuda", generator=g), dim=1).contiguous()
        seqs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
        out = ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        costs, grad, skips = out[0], out[1], out[2]
        gs = grad.double()
        print("B=%d T=%d U=%d: %.3f ms  cost %.9f  |grad| %.9f  grad.w %.9f  skips %d" % (
            B, T, U, ms, float(costs.sum()), float(gs.abs().sum()),
            float((gs * torch.arange(gs.numel(), device="cuda", dtype=torch.float64).reshape(gs.shape).remainder(7.0)).sum()),
            int(skips.sum())))


if __name__ == "__main__":
    main()
