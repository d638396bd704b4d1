#!/usr/bin/env python3
"""CTC kernels alone, float32 probabilities resident on the device (the BRNN path's I/O): the fused kernel
(ctc_fused.hip; 32-bit and float64 row store) against the three-kernel path (ctc_lattice + ctc_grad), at the
headline minibatch, the saturating batch and cfg-5's shape.  Prints one JSON line per (shape, path): wall time of
the call (host wrapper included) and GPU time between two events around it; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations (tools/profile_ctc.sh).
usage: tools/ctc_paths_bench.py [--shapes cfg3,sat,cfg5] [--reps 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
import ctc_fast  # noqa: E402

SHAPES = {"cfg3": (32, 1000, 100, 33), "sat": (4096, 1000, 100, 33), "cfg2": (256, 300, 60, 62),
          "cfg5": (8, 8000, 800, 33), "sat1k": (1024, 1000, 100, 33), "cfg4": (32, 2000, 200, 33), "sat4": (2048, 2000, 200, 33),
          # rows of 1601 states (ctc_fusedw.hip from 18 utterances on): cfg-5's minibatch and larger ones
          "cfg5x32": (32, 8000, 800, 33), "cfg5x128": (128, 8000, 800, 33)}
PATHS = {"fused": {}, "fused64": {"SCTC_CTC_STORE": "64"}, "fused2w": {"SCTC_CTC_HELPER": "0"},
         "lattice": {"SCTC_CTC_FUSED": "0"},
         # the wide fused kernel whatever the batch size (rows of 513..2048 states only)
         "wide": {"SCTC_CTC_WIDE_MIN_B": "1"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="cfg3,sat,cfg5")
    ap.add_argument("--paths", default="fused,fused64,fused2w,lattice")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    for name in args.shapes.split(","):
        B, T, U, A = SHAPES[name]
        g = torch.Generator(device="cuda")
        g.manual_seed(7)
        probs = torch.softmax(torch.randn(B * T, A, device="cuda", generator=g), dim=1)
        rs = np.random.RandomState(7)
        seqs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
        algo = B * (2 * 4 * A * T + 4 * U + 8)
        ref = None
        for pname in args.paths.split(","):
            for k in ("SCTC_CTC_STORE", "SCTC_CTC_FUSED", "SCTC_CTC_HELPER", "SCTC_CTC_WIDE_MIN_B"):
                os.environ.pop(k, None)
            os.environ.update(PATHS[pname])
            cost, grad, skip = ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B)     # warm-up (allocator, code objects)
            torch.cuda.synchronize()
            wall, gpu = [], []
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                cost, grad, skip = ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B)
                e1.record()
                torch.cuda.synchronize()
                wall.append((time.perf_counter() - t0) * 1e3)
                gpu.append(e0.elapsed_time(e1))
            c = cost.cpu().numpy()
            gsum = float(grad.double().abs().sum())
            if ref is None:
                ref = (c, grad.clone())
            out = {"shape": name, "B": B, "T": T, "U": U, "A": A, "path": pname, "wall_ms": min(wall), "gpu_ms": min(gpu),
                   "algorithmic_GBps_wall": algo / (min(wall) * 1e-3) / 1e9, "algorithmic_GBps_gpu": algo / (min(gpu) * 1e-3) / 1e9,
                   "skipped": int(skip.sum()), "cost0": float(c[0]), "sum_abs_grad": gsum,
                   "max_cost_rel_vs_first_path": float(np.max(np.abs(c - ref[0]) / np.abs(ref[0]))),
                   "max_grad_abs_vs_first_path": float((grad - ref[1]).abs().max())}
            print(json.dumps(out), flush=True)
        del probs, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
