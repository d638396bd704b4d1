#!/usr/bin/env python3
"""The units x utterances recurrent kernel (brnn_recurrent_t_kernel, more than 32 utterances; round 6) against
(a) the one-slab-per-CU kernel of rounds 1-5 (SCTC_REC_VARIANT=47: other K split, so close, not equal) and
(b) the same utterances as minibatches of 32 on the two-chain kernel, whose K split and addition order it keeps:
    hActsFor / hActsBack rows and per-utterance costs must be BIT-identical.
Prints microseconds per time step for both kernels.
usage: tools/rec_tiled_check.py [B ...]   env H (1824), T (250), RAGGED=1 (lengths ~ U[T/2, T])"""
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
D, A, NL, TL = 483, 33, 5, 3
H = int(os.environ.get("H", "1824"))
T = int(os.environ.get("T", "250"))
RAGGED = os.environ.get("RAGGED", "0") != "0"
L = _sctc.lib()


def rowbase(Ts):
    Ts = np.asarray(Ts)
    alive = np.array([(Ts > t).sum() for t in range(Ts.max())])
    return np.concatenate([[0], np.cumsum(alive)[:-1]])


def run(variant, feats, labels, Ts, timing=True, want=(100, 101, 200)):
    os.environ["SCTC_REC_VARIANT"] = variant
    B = len(Ts)
    np.random.seed(0)
    net = brnnet.NNet(D, A, H, NL, max(Ts), temporalLayer=TL, maxUtts=B)
    net.initParams()
    cost, _, skip = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
    out = {"cost": cost.copy(), "grad": net.grad.flat.clone(), "path": list(net.recurrentPath())}
    for w in want:
        out[w] = net.debugBuffer(w)
    if timing:
        L.sctc_brnn_set_profiling(net._h, 1)
        acc = np.zeros(len(PHASES))
        arr = (ctypes.c_float * len(PHASES))()
        for _ in range(3):
            net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
            L.sctc_brnn_phase_ms(net._h, arr)
            acc += np.array(list(arr))
        ph = dict(zip(PHASES, acc / 3))
        us = (ph["fwd_rec"] + ph["bwd_rec"]) * 1e3 / (2 * (max(Ts) - 1))
        out["us"] = us
        out["fwd_us"] = ph["fwd_rec"] * 1e3 / (max(Ts) - 1)
        out["bwd_us"] = ph["bwd_rec"] * 1e3 / (max(Ts) - 1)
    del net
    torch.cuda.empty_cache()
    return out


ok = True
for B in [int(v) for v in sys.argv[1:]] or [64, 128]:
    rs = np.random.RandomState(9 + B)
    Ts = sorted([int(v) for v in (rs.randint(T // 2, T + 1, size=B) if RAGGED else [T] * B)], reverse=True)
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    feats = torch.randn(sum(Ts), D, device="cuda", generator=g)
    labels = [rs.randint(1, A, size=max(t // 10, 1)).astype(np.int32) for t in Ts]
    new = run("0", feats, labels, Ts)
    old = run(os.environ.get("BASE", "47"), feats, labels, Ts)
    gd = float((new["grad"] - old["grad"]).double().norm() / old["grad"].double().norm())
    cd = float(np.abs(new["cost"] - old["cost"]).max() / np.abs(old["cost"]).max())
    fl = 2 * 2.0 * H * H * B
    line = {"H": H, "B": B, "ragged": RAGGED,
            "tiled": {"us_per_time_step": round(new["us"], 3), "fwd": round(new["fwd_us"], 3), "bwd": round(new["bwd_us"], 3),
                      "frac_of_f32_mfma_peak": round(fl / (new["us"] * 1e-6) / 1e12 / 157.3, 4), "path": new["path"]},
            "one_slab_per_cu": {"us_per_time_step": round(old["us"], 3),
                                "frac_of_f32_mfma_peak": round(fl / (old["us"] * 1e-6) / 1e12 / 157.3, 4), "path": old["path"]},
            "grad_rel_distance": gd, "cost_rel_distance": cd}
    # (b) the same utterances, 32 at a time (two-chain kernel): rows of the recurrent layer bit for bit
    rb = rowbase(Ts)
    off = np.concatenate([[0], np.cumsum(Ts)])
    same_rows = {100: True, 101: True, 200: True}
    same_cost = True
    for b0 in range(0, B, 32):
        sl = slice(b0, min(b0 + 32, B))
        Tsub = Ts[sl]
        fsub = torch.cat([feats[off[b]:off[b + 1]] for b in range(sl.start, sl.stop)])
        sub = run("0", fsub, labels[sl], Tsub, timing=False)
        same_cost = same_cost and bool((sub["cost"] == new["cost"][sl]).all())
        rbs = rowbase(Tsub)
        for w in same_rows:
            for bi, b in enumerate(range(sl.start, sl.stop)):
                tt = np.arange(Ts[b])
                if not np.array_equal(new[w][rb[tt] + b], sub[w][rbs[tt] + bi]):
                    same_rows[w] = False
    line["bit_identical_to_minibatches_of_32"] = {"costs": same_cost, "hActsFor": same_rows[100], "hActsBack": same_rows[101],
                                                  "delta_entering_layer_1": same_rows[200]}
    print(json.dumps(line), flush=True)
    ok = ok and gd < 1e-4 and cd < 1e-5 and same_cost and same_rows[100] and same_rows[101]
    del feats
sys.exit(0 if ok else 1)
