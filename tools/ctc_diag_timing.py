import os, sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/stanford-ctc_amd")
import torch, ctc_fast
B, T, U, A = 32, 1000, 100, 33
g = torch.Generator(device="cuda"); g.manual_seed(7)
probs = torch.softmax(torch.randn(B * T, A, device="cuda", generator=g), dim=1)
rs = np.random.RandomState(7)
seqs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
print("max label count per utterance:", sorted(int(np.bincount(s).max()) for s in seqs))
def t(env):
    for k in ("SCTC_CTC_DIAG", "SCTC_CTC_HELPER", "SCTC_CTC_FUSED"): os.environ.pop(k, None)
    os.environ.update(env)
    ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for name, env in [("helper full", {}), ("no phase 1", {"SCTC_CTC_DIAG": "1"}), ("helper: no finish", {"SCTC_CTC_DIAG": "2"}),
                  ("helper: no products, no finish", {"SCTC_CTC_DIAG": "6"}), ("producer does not wait", {"SCTC_CTC_DIAG": "8"}),
                  ("producer free, helper idle-ish", {"SCTC_CTC_DIAG": "14"}),
                  ("2w full", {"SCTC_CTC_HELPER": "0"}), ("2w no phase 1", {"SCTC_CTC_HELPER": "0", "SCTC_CTC_DIAG": "1"}),
                  ("lattice", {"SCTC_CTC_FUSED": "0"})]:
    print("%-34s %.3f ms (events around the call, host wrapper ~0.1 ms included)" % (name, t(env)), flush=True)
# all-distinct-ish labels: no list overflow
seqs = [((np.arange(U) % (A - 1)) + 1).astype(np.int32) for _ in range(B)]
print("cyclic labels (every label 3-4 times):")
for name, env in [("helper full", {}), ("2w full", {"SCTC_CTC_HELPER": "0"}), ("lattice", {"SCTC_CTC_FUSED": "0"})]:
    print("%-34s %.3f ms" % (name, t(env)), flush=True)
