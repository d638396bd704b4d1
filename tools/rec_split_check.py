import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/stanford-ctc_amd")
import torch
from nnets import brnnet
from oracle import brnn as obrnn
from tests.test_gpu_brnn import make_net, rel, _all_grads
H, B = 512, 48
rs = np.random.RandomState(5 * H + B)
D, A, NL, TL = 24, 33, 2, 1
Ts = [int(t) for t in rs.randint(1, 22, size=B)]
Ts[3] = 22
params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
datas = [rs.randn(D, T) for T in Ts]
labs = [rs.randint(1, A, size=max(1, T // 8)).astype(np.int32) for T in Ts]
with np.errstate(all="ignore"):
    cr, gr, sr, _ = obrnn.cost_and_grad_batch(params, datas, labs, TL)
out = {}
for eps in (0.0, 1e-5, 1e-4):
    dd = [d * (1 + eps) for d in datas]
    with np.errstate(all="ignore"):
        cr, gr, sr, _ = obrnn.cost_and_grad_batch(params, dd, labs, TL)
    for v in ("0", "45", "1"):
        os.environ["SCTC_REC_VARIANT"] = v
        net = make_net(brnnet, (D, A, H, NL, TL, max(Ts)), params, maxUtts=B)
        c, _, s = net.costAndGradBatch(dd, labs)
        g = _all_grads(net, NL)
        ref = [gr["W"][0], gr["W"][1], gr["W"][2], gr["Wf"], gr["Wb"]]
        print("eps %.0e variant %s: cost rel %.1e; grads vs oracle %s" % (eps, v, np.max(np.abs(c[~sr] - cr[~sr]) / cr[~sr]),
              ["%.1e" % rel(a, b) for a, b in zip(g, ref)]), flush=True)
        out[(eps, v)] = g
        del net
    print("   split vs single:", ["%.1e" % rel(a, b) for a, b in zip(out[(eps, "0")], out[(eps, "45")])])
