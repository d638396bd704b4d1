"""a few cfg-5 fp16 steps (T=8000, 7x2048, U=800; minibatch from argv, default 8) for profiling runs:
rocprofv3 --kernel-trace --stats -- python tools/cfg5_step.py [B] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
from nnets import brnnet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fp16 = (sys.argv[3] != "f32") if len(sys.argv) > 3 else True
D, A, H, NL, TL, T, U = 615, 33, 2048, 7, 4, 8000, 800
np.random.seed(0)
net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B, fp16=fp16)
net.initParams()
g = torch.Generator(device="cuda")
g.manual_seed(5)
feats = torch.randn(B * T, D, device="cuda", generator=g)
rs = np.random.RandomState(5)
labels = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
for _ in range(steps):
    c, _, s = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=[T] * B)
torch.cuda.synchronize()
print("cfg5 B=%d fp16=%s cost[0]=%.4f" % (B, fp16, c[0]))
