#!/usr/bin/env python3
"""The whole cfg-3 step at 128 utterances of T = 1000 (HBM-resident features), frames/s: the default recurrence and
SCTC_REC_VARIANT=47 (round 5's one-slab-per-CU kernel) in alternation.  usage: tools/minibatch128_step.py [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
from nnets import brnnet  # noqa: E402

D, A, H, NL, TL, T, B = 483, 33, 1824, 5, 3, 1000, 128
g = torch.Generator(device="cuda")
g.manual_seed(9)
feats = torch.randn(B * T, D, device="cuda", generator=g)
rs = np.random.RandomState(9)
labels = [rs.randint(1, A, size=T // 10).astype(np.int32) for _ in range(B)]
Ts = [T] * B
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for variant in ("0", "47"):
        os.environ["SCTC_REC_VARIANT"] = variant
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
        net.initParams()
        net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("variant %s: %.1f ms per step, %.0f frames/s" % (variant, dt * 1e3, B * T / dt), flush=True)
        del net
        torch.cuda.empty_cache()
