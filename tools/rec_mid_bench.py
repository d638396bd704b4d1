#!/usr/bin/env python3
"""The recurrent time step at 6..16 utterances: the sentinel / MFMA kernel (default) against the single-chain flag
kernel (the default since round 5; SCTC_REC_VARIANT=42 = the sentinel kernel), cfg-3 layer sizes; microseconds per time step, bit-identity of costs, gradient distance.
usage: tools/rec_mid_bench.py [B ...]"""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch, _sctc
from nnets import brnnet
PHASES = ["fwd_gemm", "fwd_rec", "ctc", "bwd_gemm", "bwd_rec", "other"]
D, A, H, NL, TL, T = 483, 33, int(os.environ.get("H", "1824")), 5, 3, 250
L = _sctc.lib()
for B in [int(v) for v in sys.argv[1:]] or [6, 8, 12, 16]:
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    feats = torch.randn(B * T, D, device="cuda", generator=g)
    rs = np.random.RandomState(9)
    labels = [rs.randint(1, A, size=T // 10).astype(np.int32) for _ in range(B)]
    Ts = [T] * B
    res, grads = {}, {}
    for variant in ("42", "0"):
        os.environ["SCTC_REC_VARIANT"] = variant
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
        net.initParams()
        cost, _, _ = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
        grads[variant] = (cost.copy(), net.grad.flat.clone())
        L.sctc_brnn_set_profiling(net._h, 1)
        acc = np.zeros(len(PHASES)); arr = (ctypes.c_float * len(PHASES))()
        for _ in range(3):
            net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
            L.sctc_brnn_phase_ms(net._h, arr); acc += np.array(list(arr))
        ph = dict(zip(PHASES, acc / 3))
        res[variant] = {"us_per_time_step": round((ph["fwd_rec"] + ph["bwd_rec"]) * 1e3 / (2 * (T - 1)), 3), "path": list(net.recurrentPath())}
        del net; torch.cuda.empty_cache()
    gd = float((grads["42"][1] - grads["0"][1]).double().norm() / grads["42"][1].double().norm())
    cd = float(np.max(np.abs(grads["42"][0] - grads["0"][0]) / np.abs(grads["42"][0])))
    print(json.dumps({"H": H, "B": B, "sentinel_mfma": res["42"], "flag_single_chain": res["0"], "cost_rel_diff": cd, "grad_rel_diff": gd}), flush=True)
