#!/usr/bin/env python3
"""Per-phase timeline of brnn_recurrent_t_kernel (SCTC_REC_DEBUG=1 stamps, 100 MHz wall clock): steps 64..71 of both
sub-chains on the first and last unit block of combo 0, forward pass and BPTT.
usage: tools/build_variant.sh stamps recurrent.hip -DSCTC_T_STAMPS      (or -DSCTC_T_STAMP_CHUNKS: one stamp per chunk as well)
       SCTC_LIB_PATH=stanford-ctc_amd/libvar_stamps.so tools/rec_tiled_timeline.py [B ...]   env H (1824)
(the production library has no stamps in this kernel: eight branches per phase cost it 0.1-0.2 us per phase)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
os.environ["SCTC_REC_DEBUG"] = "1"
import torch  # noqa: E402
import _sctc  # noqa: E402
from nnets import brnnet  # noqa: E402

D, A, NL, TL, T = 483, 33, 5, 3, 200
H = int(os.environ.get("H", "1824"))
NAMES = ["setup + chunk 0", "chunk 1 (+ epilogue pieces)", "chunk 2", "chunk 3", "chunk 4", "up to the publish + late loads",
         "remaining batches"]
for B in [int(v) for v in sys.argv[1:]] or [64, 128]:
    np.random.seed(0)
    net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
    net.initParams()
    rs = np.random.RandomState(1)
    feats = torch.randn(B * T, D, device="cuda")
    labels = [rs.randint(1, A, size=20).astype(np.int32) for _ in range(B)]
    for _ in range(2):
        net.costAndGradBatch(None, labels, feats_dev=feats, T_b=[T] * B)
    W = 2 * 16 * 8 + 512 * 8
    buf = np.zeros(2 * W, dtype=np.uint32)
    _sctc.lib().sctc_brnn_debug_read(net._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    st = buf.reshape(2, W)[:, :256].reshape(2, 2, 16, 8).astype(np.int64)   # [pass][wg][phase slot][stamp]
    for ps, pname in enumerate(("forward", "bptt")):
        for w in range(2):
            s = st[ps, w]
            d = np.diff(s, axis=1) * 0.01                 # us between consecutive stamps of a phase
            per_phase = np.diff(s[:, 0]) * 0.01           # phase start -> next phase start
            print("B=%d %s wg %s: phase-to-phase %.2f us (A->B %.2f, B->A %.2f) => %.2f us per time step" % (
                B, pname, "first" if w == 0 else "last", np.median(per_phase), np.median(per_phase[0::2]),
                np.median(per_phase[1::2]), np.median(per_phase[0::2]) + np.median(per_phase[1::2])))
            print("     " + "; ".join("%s %.2f" % (n, v) for n, v in zip(NAMES, np.median(d, axis=0))) +
                  "; partial sums + next setup start %.2f" % (np.median(per_phase) - np.median(d, axis=0).sum()))
    fine = buf.reshape(2, W)[:, 256:256 + 2 * 8 * 32].reshape(2, 2, 8, 32).astype(np.int64)    # [pass][wg][phase][chunk stamp]
    for ps, pname in enumerate(("forward", "bptt")):
        f = fine[ps, 0]
        n = int((f[0] != 0).sum())
        if n < 4:         # the library was not built with -DSCTC_T_STAMP_CHUNKS
            continue
        d = np.diff(f[:, :n], axis=1) * 0.01
        print("B=%d %s wg first, us per chunk (median of 8 phases), chunk 0 includes the phase's setup:" % (B, pname))
        print("     " + " ".join("%.2f" % v for v in np.median(d, axis=0)))
    del net, feats
    torch.cuda.empty_cache()
