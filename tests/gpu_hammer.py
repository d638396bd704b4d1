"""One process of the shared-device hammer test (tests/test_gpu_shared.py): runs `steps`
costAndGradBatch calls of a B-utterance minibatch at H = 1824 (every call launches two whole-device
persistent recurrent grids) and prints one JSON line with a digest of every step's costs and
gradients.  Two of these started together on ONE GPU must print what a solo run prints.

usage: gpu_hammer.py <steps> <B> <T> <sync_dir> <tag>
A file <sync_dir>/<tag>.ready is written after warm-up; the timed loop starts once
<sync_dir>/go exists (the test writes it when every process is ready)."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    steps, B, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    sync_dir, tag = sys.argv[4], sys.argv[5]
    import torch
    import _sctc
    from nnets import brnnet
    D, A, H, NL, TL = 40, 33, 1824, 3, 2
    np.random.seed(77)
    net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
    net.initParams()
    rs = np.random.RandomState(5)
    datas = [rs.randn(D, T - (b % 3)).astype(np.float32) for b in range(B)]
    labs = [rs.randint(1, A, size=max(1, T // 10)).astype(np.int32) for _ in range(B)]
    net.costAndGradBatch(datas, labs)                # warm-up: code objects, workspace touch
    torch.cuda.synchronize()
    open(os.path.join(sync_dir, tag + ".ready"), "w").close()
    t0 = time.time()
    while not os.path.exists(os.path.join(sync_dir, "go")):
        if time.time() - t0 > 600:
            raise SystemExit("peer never became ready")
        time.sleep(0.005)
    digests, costs_all = [], []
    t0 = time.time()
    for k in range(steps):
        costs, _, skips = net.costAndGradBatch(datas, labs)
        g = net.grad.flat.cpu().numpy()
        digests.append(hashlib.sha1(costs.tobytes() + g.tobytes()).hexdigest()[:16])
        costs_all.append([float(c) for c in costs])
        # every step moves the weights a little so that consecutive steps differ
        net.updateParams(-1e-5, net.grad)
    dt = time.time() - t0
    fwd, bptt, retries = net.recurrentPath()
    print(json.dumps({"tag": tag, "digests": digests, "costs": costs_all, "seconds": dt,
                      "path": [fwd, bptt], "retries": retries,
                      "shared_mode": int(_sctc.lib().sctc_shared_device())}))


if __name__ == "__main__":
    main()
