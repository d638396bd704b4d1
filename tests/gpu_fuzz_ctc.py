"""Randomised differential test of the CTC kernels against the C oracle (`python tests/gpu_fuzz_ctc.py [n_cases] [seed]`
on the GPU; tests/test_gpu_fuzz.py runs a 20-case subset with a fixed seed in the suite): random alphabet, ragged
batches, repeats, labels equal to the blank, T < U (empty band), infeasible repeats and
zero-probability labels (skip), peaked and flat distributions, float32 and float64 I/O,
one- and four-wave lattices; round 6: rows of up to 2001 states in batches of up to 20 (the wide fused kernel from 18 / 24
utterances on; `SCTC_CTC_WIDE=1 python tests/gpu_fuzz_ctc.py ...` soaks it on every shape)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402,F401
import ctc_fast  # noqa: E402
from oracle import ctc as octc  # noqa: E402


def run(n_cases=60, seed=0):
    rs = np.random.RandomState(seed)
    worst_c = worst_g = 0.0
    n_skip = n_inf = 0
    for case in range(n_cases):
        A = int(rs.choice([3, 5, 28, 33, 62, 100, 200]))
        B = int(rs.choice([1, 2, 5, 9, 13, 20, 32]))
        blank = int(rs.choice([0, 0, 0, A - 1, A // 2]))
        long_rows = rs.rand() < 0.25          # > 256 lattice states: the four-wave kernel
        very_long = B <= 20 and rs.rand() < 0.2   # up to 2001 states: four / eight waves, the wide fused kernel from 18 / 24 utterances on
        peaked = rs.rand() < 0.5
        f64 = rs.rand() < 0.4
        probs, seqs = [], []
        for b in range(B):
            if very_long:
                U = int(rs.randint(200, 1001))
                T = int(rs.randint(U, 2 * U + 40))
            elif long_rows:
                U = int(rs.randint(130, 420))
                T = int(rs.randint(U, 3 * U))
            else:
                U = int(rs.randint(1, 60))
                T = int(rs.randint(1, 200))
            kind = rs.rand()
            if kind < 0.1:
                T = max(1, U - int(rs.randint(1, 4)))                    # T < U: empty band
            seq = rs.randint(0, A, size=U).astype(np.int32)              # may contain the blank id
            if kind > 0.9 and U >= 2:
                seq[:] = seq[0]                                          # all repeats: often infeasible
            logits = rs.randn(A, T) * (6.0 if peaked else 1.0)
            p = np.exp(logits - logits.max(axis=0))
            p /= p.sum(axis=0)
            if 0.1 <= kind < 0.15:
                p[seq[0], :] = 0.0                                       # zero-probability label
            probs.append(np.asfortranarray(p if f64 else p.astype(np.float32)))
            seqs.append(seq)
        with np.errstate(all="ignore"):
            cost, grads, skip = ctc_fast.ctc_loss_batch(probs, seqs, blank=blank)
        for b in range(B):
            with np.errstate(all="ignore"):
                c_ref, g_ref, s_ref = octc.ctc_loss(np.asfortranarray(probs[b].astype(np.float64)), seqs[b], blank)
            assert bool(skip[b]) == bool(s_ref), (case, b, "skip", skip[b], s_ref)
            if s_ref:
                n_skip += 1
                continue
            if np.isinf(c_ref):
                n_inf += 1
                assert np.isinf(cost[b]) and cost[b] > 0, (case, b, cost[b])
                continue
            dc = abs(cost[b] - c_ref) / max(abs(c_ref), 1e-30)
            dg = np.max(np.abs(grads[b].astype(np.float64) - g_ref))
            tol_c, tol_g = (1e-10, 1e-8) if f64 else (1e-5, 2e-5)
            assert dc < tol_c and dg < tol_g, (case, b, A, probs[b].shape, len(seqs[b]), f64, dc, dg)
            worst_c, worst_g = max(worst_c, dc if f64 else 0.0), max(worst_g, dg if f64 else 0.0)
        print("case %2d A=%3d B=%2d blank=%3d long=%d peaked=%d f64=%d ok" % (case, A, B, blank, 2 if very_long else long_rows, peaked, f64), flush=True)
    print("all %d cases agree (%d skipped utterances, %d empty bands; float64 worst: cost %.1e grad %.1e)"
          % (n_cases, n_skip, n_inf, worst_c, worst_g))


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
