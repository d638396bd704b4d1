#!/usr/bin/env python3
"""us per time step of the recurrence for given SCTC_REC_VARIANT values (diagnostic variants included: results are not
checked).  usage: tools/rec_variant_time.py B variant [variant ...]   env H (1824), T (250)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
import _sctc  # noqa: E402
from nnets import brnnet  # noqa: E402

PHASES = ["fwd_gemm", "fwd_rec", "ctc", "bwd_gemm", "bwd_rec", "other"]
D, A, NL, TL = 483, 33, 5, 3
H, T = int(os.environ.get("H", "1824")), int(os.environ.get("T", "250"))
B = int(sys.argv[1])
L = _sctc.lib()
feats = torch.randn(B * T, D, device="cuda")
rs = np.random.RandomState(9)
labels = [rs.randint(1, A, size=T // 10).astype(np.int32) for _ in range(B)]
for rep in range(2):
    for variant in sys.argv[2:]:
        os.environ["SCTC_REC_VARIANT"] = variant
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
        net.initParams()
        net.costAndGradBatch(None, labels, feats_dev=feats, T_b=[T] * B)
        L.sctc_brnn_set_profiling(net._h, 1)
        acc = np.zeros(len(PHASES))
        arr = (ctypes.c_float * len(PHASES))()
        for _ in range(3):
            net.costAndGradBatch(None, labels, feats_dev=feats, T_b=[T] * B)
            L.sctc_brnn_phase_ms(net._h, arr)
            acc += np.array(list(arr))
        ph = dict(zip(PHASES, acc / 3))
        print("B=%d variant %s: %.2f us per time step (fwd %.2f, bwd %.2f)" % (
            B, variant, (ph["fwd_rec"] + ph["bwd_rec"]) * 1e3 / (2 * (T - 1)), ph["fwd_rec"] * 1e3 / (T - 1), ph["bwd_rec"] * 1e3 / (T - 1)), flush=True)
        del net
        torch.cuda.empty_cache()
