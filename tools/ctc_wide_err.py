#!/usr/bin/env python3
"""Per utterance: worst |grad - oracle| of the wide fused CTC kernel with 32-bit rows, with float64 rows and of the lattice + grad
kernels at the cfg-5 shape (T = 8000, U = 800), and where (frame, symbol) it occurs -- the frames around t = U and t = T - U,
where the reference's own alpha-beta overlap is a float64 denormal.  usage: tools/ctc_wide_err.py [B]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
import ctc_fast  # noqa: E402
from oracle import ctc as octc  # noqa: E402
A, T, U, B = 33, 8000, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device="cuda"); g.manual_seed(7)
probs = torch.softmax(torch.randn(B * T, A, device="cuda", generator=g), dim=1)
rs = np.random.RandomState(7)
seqs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
res = {}
for name, env in (("wide32", {"SCTC_CTC_WIDE_MIN_B": "1"}), ("wide64", {"SCTC_CTC_WIDE_MIN_B": "1", "SCTC_CTC_STORE": "64"}), ("lattice", {"SCTC_CTC_WIDE": "0"})):
    for k in ("SCTC_CTC_WIDE_MIN_B", "SCTC_CTC_STORE", "SCTC_CTC_WIDE"): os.environ.pop(k, None)
    os.environ.update(env)
    for rep in range(2):
        c, gr, s = ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B)
        res[name + str(rep)] = gr.clone()
ph = probs.cpu().numpy().astype(np.float64)
worst = []
for b in range(B):
    y = np.asfortranarray(ph[b * T:(b + 1) * T].T)
    c_ref, g_ref, s_ref = octc.ctc_loss(y, seqs[b])
    row = [b]
    for name in ("wide320", "wide321", "wide640", "lattice0"):
        gg = res[name][b * T:(b + 1) * T].cpu().numpy().astype(np.float64).T
        e = np.abs(gg - g_ref)
        k, t = np.unravel_index(e.argmax(), e.shape)
        row.append("%s %.1e@t=%d,k=%d(y=%.1e,g=%.2e)" % (name, e.max(), t, k, y[k, t], g_ref[k, t]))
    print(*row, flush=True)
print("run-to-run wide32:", float((res["wide320"] - res["wide321"]).abs().max()))
