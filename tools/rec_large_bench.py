#!/usr/bin/env python3
"""The recurrent time step at more than 32 utterances (brnn_recurrent_kernel<NTW>): microseconds per time step and
fraction of the fp32 MFMA peak for the default kernel (exchange loads of batch k+1 under the MFMAs of batch k) and
SCTC_REC_VARIANT=40 (rounds 1-4: loads, fence, MFMAs), cfg-3 layer sizes, and a bit-identity check of the two.
usage: tools/rec_large_bench.py [B ...]    (default 48 64 96 128)"""
import ctypes
import json
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
D, A, H, NL, TL, T = 483, 33, int(os.environ.get("H", "1824")), 5, 3, 250
L = _sctc.lib()
for B in [int(v) for v in sys.argv[1:]] or [48, 64, 96, 128]:
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    feats = torch.randn(B * T, D, device="cuda", generator=g)
    rs = np.random.RandomState(9)
    labels = [rs.randint(1, A, size=T // 10).astype(np.int32) for _ in range(B)]
    Ts = [T] * B
    res, grads = {}, {}
    for variant in ("0", os.environ.get("REC_BASE_VARIANT", "40")):
        os.environ["SCTC_REC_VARIANT"] = variant
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
        net.initParams()
        cost, _, _ = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
        grads[variant] = (cost.copy(), net.grad.flat.clone())
        L.sctc_brnn_set_profiling(net._h, 1)
        acc = np.zeros(len(PHASES))
        arr = (ctypes.c_float * len(PHASES))()
        for _ in range(3):
            net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
            L.sctc_brnn_phase_ms(net._h, arr)
            acc += np.array(list(arr))
        ph = dict(zip(PHASES, acc / 3))
        us = (ph["fwd_rec"] + ph["bwd_rec"]) * 1e3 / (2 * (T - 1))
        fl = 2 * 2.0 * H * H * B
        res[variant] = {"us_per_time_step": round(us, 3), "frac_of_f32_mfma_peak": round(fl / (us * 1e-6) / 1e12 / 157.3, 4),
                        "path": list(net.recurrentPath())}
        del net
        torch.cuda.empty_cache()
    same = bool((grads["0"][0] == grads[os.environ.get("REC_BASE_VARIANT", "40")][0]).all() and torch.equal(grads["0"][1], grads[os.environ.get("REC_BASE_VARIANT", "40")][1]))
    print(json.dumps({"H": H, "B": B, "pipelined": res["0"], "rounds_1_4": res[os.environ.get("REC_BASE_VARIANT", "40")], "bit_identical": same}), flush=True)
    base = os.environ.get("REC_BASE_VARIANT", "40")
    gd = float((grads["0"][1] - grads[base][1]).double().norm() / grads[base][1].double().norm())
    print("   gradient relative distance %.1e" % gd, flush=True)
    assert same or (base != "40" and gd < 1e-4)       # other kernels: other summation order
    del feats
