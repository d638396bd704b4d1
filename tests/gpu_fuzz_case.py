"""Replays ONE case of tests/gpu_fuzz.py (`python tests/gpu_fuzz_case.py SEED CASE`) with both recurrent
kernel variants and the float64 oracle at any layer size: which side of a disagreement is wrong?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
from nnets import brnnet  # noqa: E402
from oracle import brnn as obrnn  # noqa: E402
from tests.gpu_fuzz import grads, host_stack, rel  # noqa: E402


def main(seed, want):
    rs = np.random.RandomState(seed)
    for case in range(want + 1):
        H = int(rs.choice([512, 512, 1024, 1824, 2048, 96]))
        B = int(rs.choice([1, 2, 3, 4, 5, 6, 8, 11, 16, 17, 24, 32, 33, 40]))
        NL = int(rs.choice([2, 3]))
        TL = int(rs.randint(1, NL))
        D, A = 24, int(rs.choice([33, 62]))
        Tmax = int(rs.randint(2, 26))
        Ts = [int(t) for t in rs.randint(1, Tmax + 1, size=B)]
        Ts[int(rs.randint(B))] = Tmax
        params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
        reg = float(rs.choice([0.0, 0.0, 1e-3]))
        max_act = float(rs.choice([20.0, 20.0, 3.0, 0.5]))
        scale_w = float(rs.choice([1.0, 1.0, 0.05]))
        if scale_w != 1.0:
            params["W"] = [w * scale_w for w in params["W"]]
            params["Wf"] = params["Wf"] * scale_w
            params["Wb"] = params["Wb"] * scale_w
        datas = [rs.randn(D, T) for T in Ts]
        labs = [rs.randint(0, A, size=max(1, T // 6)).astype(np.int32) for T in Ts]
    print("case %d: H=%d B=%d NL=%d TL=%d A=%d Tmax=%d Ts=%s reg=%g maxAct=%g w*%g" % (want, H, B, NL, TL, A, Tmax, Ts, reg, max_act, scale_w))
    with np.errstate(all="ignore"):
        cr, gr, sr, _ = obrnn.cost_and_grad_batch(params, datas, labs, TL, max_act=max_act, reg=reg)
    ref = gr["W"] + [gr["Wf"], gr["Wb"]]
    names = ["W%d" % (i + 1) for i in range(NL + 1)] + ["Wf", "Wb"]
    res = {}
    for variant in ("0", "1"):
        os.environ["SCTC_REC_VARIANT"] = variant
        net = brnnet.NNet(D, A, H, NL, Tmax, temporalLayer=TL, maxUtts=B, reg=reg)
        net.maxAct = max_act
        net.setParams(host_stack(params))
        costs, _, skips = net.costAndGradBatch(datas, labs)
        g = grads(net, NL)
        res[variant] = g
        ok = ~skips
        print("variant %s: skips %s oracle %s | cost err %.1e | grad err vs oracle:" % (
            variant, skips.astype(int).tolist(), sr.astype(int).tolist(),
            np.max(np.abs(costs[ok] - cr[ok]) / np.abs(cr[ok]))), {n: "%.1e" % rel(a, b) for n, a, b in zip(names, g, ref)})
        del net
    print("variant 0 vs 1:", {n: "%.1e" % rel(a, b) for n, a, b in zip(names, res["0"], res["1"])})
    # a unit within fp32 rounding of the clip boundary gets its mask from the summation order (a
    # "boundary flip", ~1e-3 of the gradient each); a real defect would not come and go under tiny
    # perturbations of the inputs
    for eps in (1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4):
        d2 = [d * (1.0 + eps) for d in datas]
        with np.errstate(all="ignore"):
            cr2, gr2, sr2, _ = obrnn.cost_and_grad_batch(params, d2, labs, TL, max_act=max_act, reg=reg)
        ref2 = gr2["W"] + [gr2["Wf"], gr2["Wb"]]
        line = []
        for variant in ("0", "1"):
            os.environ["SCTC_REC_VARIANT"] = variant
            net = brnnet.NNet(D, A, H, NL, Tmax, temporalLayer=TL, maxUtts=B, reg=reg)
            net.maxAct = max_act
            net.setParams(host_stack(params))
            net.costAndGradBatch(d2, labs)
            g = grads(net, NL)
            line.append("variant %s worst %.1e" % (variant, max(rel(a, b) for a, b in zip(g, ref2))))
            del net
        print("inputs * (1 + %.0e): %s" % (eps, "; ".join(line)))


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
