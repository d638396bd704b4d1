"""Hunts for the condition under which the one-workgroup-per-CU flag kernel (SCTC_REC_VARIANT=1)
disagrees with the oracle: small layer, many (B, lengths) patterns."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
from nnets import brnnet  # noqa: E402
from oracle import brnn as obrnn  # noqa: E402
from tests.gpu_fuzz import grads, host_stack, rel  # noqa: E402


def check(H, Ts, NL=3, TL=1, seed=0, variant="1"):
    rs = np.random.RandomState(seed)
    D, A = 24, 33
    B, Tmax = len(Ts), max(Ts)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    datas = [rs.randn(D, T) for T in Ts]
    labs = [rs.randint(0, A, size=max(1, T // 6)).astype(np.int32) for T in Ts]
    with np.errstate(all="ignore"):
        cr, gr, sr, _ = obrnn.cost_and_grad_batch(params, datas, labs, TL, max_act=20.0)
    ref = gr["W"] + [gr["Wf"], gr["Wb"]]
    os.environ["SCTC_REC_VARIANT"] = variant
    net = brnnet.NNet(D, A, H, NL, Tmax, temporalLayer=TL, maxUtts=B)
    net.setParams(host_stack(params))
    costs, _, skips = net.costAndGradBatch(datas, labs)
    g = grads(net, NL)
    del net
    names = ["W%d" % (i + 1) for i in range(NL + 1)] + ["Wf", "Wb"]
    errs = {n: rel(a, b) for n, a, b in zip(names, g, ref)}
    return max(errs.values()), errs


if __name__ == "__main__":
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    pats = {
        "17 equal T=12": [12] * 17,
        "17, last short": [12] * 16 + [5],
        "17, last T=1": [12] * 16 + [1],
        "17, one T=1 inside": [12] * 8 + [1] + [12] * 8,
        "17 fuzz lengths": [17, 3, 1, 16, 18, 6, 11, 3, 9, 13, 8, 18, 4, 9, 5, 14, 8],
        "16 fuzz lengths": [17, 3, 1, 16, 18, 6, 11, 3, 9, 13, 8, 18, 4, 9, 5, 14],
        "17 fuzz w/o T=1": [17, 3, 2, 16, 18, 6, 11, 3, 9, 13, 8, 18, 4, 9, 5, 14, 8],
        "33 ragged": list(np.random.RandomState(1).randint(1, 20, size=33)),
        "40 ragged": list(np.random.RandomState(2).randint(2, 20, size=40)),
        "20 ragged": list(np.random.RandomState(3).randint(1, 20, size=20)),
    }
    for name, Ts in pats.items():
        Ts = [int(t) for t in Ts]
        for variant in ("1", "0"):
            worst, errs = check(H, Ts, variant=variant)
            flag = "  <-- WRONG" if worst > 1e-4 else ""
            print("H=%d %-22s variant %s: worst %.1e %s%s" % (H, name, variant, worst,
                  {k: "%.0e" % v for k, v in errs.items() if v > 1e-4}, flag))
