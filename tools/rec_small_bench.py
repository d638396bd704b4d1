import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/stanford-ctc_amd")
import torch, _sctc
from nnets import brnnet
PHASES = ["fwd_gemm", "fwd_rec", "ctc", "bwd_gemm", "bwd_rec", "other"]
D, A, NL, TL, T = 483, 33, 5, 3, 250
L = _sctc.lib()
for H in (1824, 2048, 1024, 512):
  for B in (1, 2, 3, 4, 5):
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    feats = torch.randn(B * T, D, device="cuda", generator=g)
    rs = np.random.RandomState(9)
    labels = [rs.randint(1, A, size=T // 10).astype(np.int32) for _ in range(B)]
    res = {}; gr = {}
    for variant in ("0", "44"):
        os.environ["SCTC_REC_VARIANT"] = variant
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B); net.initParams()
        cost, _, _ = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=[T] * B)
        gr[variant] = (cost.copy(), net.grad.flat.clone())
        L.sctc_brnn_set_profiling(net._h, 1)
        acc = np.zeros(6); arr = (ctypes.c_float * 6)()
        for _ in range(3):
            net.costAndGradBatch(None, labels, feats_dev=feats, T_b=[T] * B); L.sctc_brnn_phase_ms(net._h, arr); acc += np.array(list(arr))
        ph = dict(zip(PHASES, acc / 3))
        res[variant] = round((ph["fwd_rec"] + ph["bwd_rec"]) * 1e3 / (2 * (T - 1)), 3)
        del net; torch.cuda.empty_cache()
    gd = float((gr["0"][1] - gr["44"][1]).double().norm() / gr["0"][1].double().norm())
    print(json.dumps({"H": H, "B": B, "default_us": res["0"], "q1_us": res["44"], "grad_rel_diff": gd}), flush=True)
