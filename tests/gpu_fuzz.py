"""Randomised differential test of the recurrent kernels (`python tests/gpu_fuzz.py [n_cases] [seed]` on the GPU; tests/test_gpu_fuzz.py runs a 20-case
subset with a fixed seed in the suite): random layer size, minibatch size,
ragged lengths and time order; the automatically chosen kernel (sentinel/VALU, sentinel/MFMA,
two-chain, flag) against the one-workgroup-per-CU flag kernel (SCTC_REC_VARIANT=1) and, for the
small layer sizes, the float64 oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
from nnets import brnnet  # noqa: E402
from oracle import brnn as obrnn  # noqa: E402


def rel(a, b):
    return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-30)


def host_stack(params):
    st = [[w, b] for w, b in zip(params["W"], params["b"])]
    st += [[params["Wf"], None], [params["Wb"], None]]
    return st


def grads(net, NL):
    return [net.grad[i][0].copy_to_host().astype(np.float64).copy() for i in range(NL + 3)]


def run(n_cases=40, seed=0):
    rs = np.random.RandomState(seed)
    worst = 0.0
    for case in range(n_cases):
        H = int(rs.choice([512, 512, 1024, 1824, 2048, 96]))
        B = int(rs.choice([1, 2, 3, 4, 5, 6, 8, 11, 16, 17, 24, 32, 33, 40, 48, 64, 65, 81, 100, 128]))   # (round 6: beyond 40 -- the tiled kernel, the launch cuts)
        NL = int(rs.choice([2, 3]))
        TL = int(rs.randint(1, NL))
        D, A = 24, int(rs.choice([33, 62]))
        Tmax = int(rs.randint(2, 26))
        Ts = [int(t) for t in rs.randint(1, Tmax + 1, size=B)]
        Ts[int(rs.randint(B))] = Tmax
        params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
        reg = float(rs.choice([0.0, 0.0, 1e-3]))
        max_act = float(rs.choice([20.0, 20.0, 3.0, 0.5]))
        scale_w = float(rs.choice([1.0, 1.0, 0.05]))     # small weights: unsaturated units, gradients through every path
        if scale_w != 1.0:
            for key in ("W",):
                params[key] = [w * scale_w for w in params[key]]
            params["Wf"] = params["Wf"] * scale_w
            params["Wb"] = params["Wb"] * scale_w
        datas = [rs.randn(D, T) for T in Ts]
        labs = [rs.randint(0, A, size=max(1, T // 6)).astype(np.int32) for T in Ts]
        def both(data_list):
            res = []
            for variant in (os.environ.get("SCTC_FUZZ_VARIANT", "0"), "1"):   # SCTC_FUZZ_VARIANT: the kernel under test (default: the automatic choice)
                os.environ["SCTC_REC_VARIANT"] = variant
                net = brnnet.NNet(D, A, H, NL, Tmax, temporalLayer=TL, maxUtts=B, reg=reg)
                net.maxAct = max_act
                net.setParams(host_stack(params))
                costs, _, skips = net.costAndGradBatch(data_list, labs)
                res.append((costs.copy(), skips.copy(), grads(net, NL)))
                del net
            (c0, s0, g0), (c1, s1, g1) = res
            assert (s0 == s1).all(), (case, H, B)
            ok = ~s0
            dc = np.max(np.abs(c0[ok] - c1[ok]) / np.maximum(np.abs(c1[ok]), 1e-30)) if ok.any() else 0.0
            return c0, s0, g0, dc, max(rel(a, b) for a, b in zip(g0, g1))

        c0, s0, g0, dc, dg = both(datas)
        note = ""
        if dg >= 1e-4:
            # The reference's init saturates most units at the [0,20] clip; a unit that sits within
            # fp32 rounding of a boundary gets its mask from the summation order, and the few
            # unsaturated units carry the whole gradient.  Such a case must disappear under a 1e-5
            # relative perturbation of the inputs (a real defect would not).
            # (one perturbation is not enough: the flip can move to the other variant or to another
            # unit -- seed 13 case 50 kept 4.7e-3 at 1e-5 and lost it at 1e-4, tests/gpu_fuzz_case.py)
            note = " (boundary flip at %.1e, re-run perturbed)" % dg
            base = datas
            for eps in (1e-5, 1e-4, 3e-4, 1e-3):
                datas = [d * (1.0 + eps) for d in base]
                c0, s0, g0, dc, dg = both(datas)
                if dg < 1e-4:
                    break
        msg = "case %2d H=%4d B=%2d NL=%d TL=%d A=%d Tmax=%2d reg=%g maxAct=%g w*%g skipped=%d: cost %.1e grad %.1e%s" % (
            case, H, B, NL, TL, A, Tmax, reg, max_act, scale_w, int(s0.sum()), dc, dg, note)
        if H <= 512:
            with np.errstate(all="ignore"):
                cr, gr, sr, _ = obrnn.cost_and_grad_batch(params, datas, labs, TL, max_act=max_act, reg=reg)
            assert (sr == s0).all(), (case, "skip vs oracle")
            do = rel(g0[0], gr["W"][0])
            msg += " | vs oracle %.1e" % do
            # fp32 vs fp64 can put a unit on the other side of the clip boundary (more likely with a
            # low ceiling); a handful of such flips moves the gradient by up to ~1e-2
            assert do < (2e-3 if max_act >= 20.0 else 2e-2), msg
        print(msg, flush=True)
        assert dc < 1e-5 and dg < (1e-4 if max_act >= 20.0 else 2e-2), msg
        worst = max(worst, dg)
    print("all %d cases agree (worst gradient difference %.1e)" % (n_cases, worst))


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
