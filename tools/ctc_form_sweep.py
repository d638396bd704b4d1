#!/usr/bin/env python3
"""Which form of the narrow fused CTC kernel at which batch size (cfg-3 shape, float32 probabilities on the device):
helper waves (six waves per utterance), two waves, two waves on the register diet -- ms between events around the Python
entry, best of 5.  The dispatch's thresholds (SCTC_CTC_HELPER_MAX_B = 256, SCTC_CTC_DIET_MIN_B = 1025) come from this table.
usage: tools/ctc_form_sweep.py [B ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
import ctc_fast  # noqa: E402

A, T, U = 33, 1000, 100
FORMS = {"helper": {"SCTC_CTC_HELPER": "1"}, "two-wave": {"SCTC_CTC_HELPER": "0", "SCTC_CTC_DIET_MIN_B": "1000000"},
         "two-wave, diet": {"SCTC_CTC_HELPER": "0", "SCTC_CTC_DIET_MIN_B": "1"}}
for B in [int(v) for v in sys.argv[1:]] or [64, 128, 256, 384, 512, 1024, 1536, 2048, 4096]:
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    probs = torch.softmax(torch.randn(B * T, A, device="cuda", generator=g), dim=1)
    rs = np.random.RandomState(7)
    seqs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    row = []
    for name, env in FORMS.items():
        for k in ("SCTC_CTC_HELPER", "SCTC_CTC_DIET_MIN_B"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row.append("%s %.3f" % (name, best))
    print("B=%5d  " % B + " | ".join(row), flush=True)
    del probs
    torch.cuda.empty_cache()
