"""Randomised test of sctc_gemm_f32 (`python tests/gpu_fuzz_gemm.py [n_cases] [seed]` on the GPU;
tests/test_gpu_fuzz.py runs a subset with a fixed seed in the suite): random sizes (multiples of 4, ragged against every tile shape),
all four operand layouts, padded leading dimensions, bias/ReLU epilogue, with and without split-K
workspace, against a float64 product."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
import _sctc  # noqa: E402


def run(n_cases=100, seed=0):
    rs = np.random.RandomState(seed)
    L = _sctc.lib()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    worst = 0.0
    for case in range(n_cases):
        M = 4 * int(rs.choice([1, 2, 7, 16, 24, 25, 31, 32, 33, 57, 100, 250, 456, 512]))
        N = 4 * int(rs.choice([1, 3, 8, 16, 24, 25, 32, 48, 57, 120, 456]))
        K = 4 * int(rs.choice([1, 2, 3, 4, 5, 9, 16, 64, 129, 456, 1000]))
        akc, bkc = int(rs.randint(2)), int(rs.randint(2))
        pad_a, pad_b, pad_c = (4 * int(rs.randint(0, 5)) for _ in range(3))
        a = torch.randn((M, K + pad_a) if akc else (K, M + pad_a), device="cuda")
        b = torch.randn((N, K + pad_b) if bkc else (K, N + pad_b), device="cuda")
        c = torch.full((M, N + pad_c), 7.0, device="cuda")
        bias = torch.randn(N, device="cuda") if rs.rand() < 0.5 else None
        relu = int(rs.rand() < 0.5)
        use_ws = rs.rand() < 0.7
        rc = L.sctc_gemm_f32(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc, c.data_ptr(),
                             c.shape[1], M, N, K, bias.data_ptr() if bias is not None else None, relu,
                             ws.data_ptr() if use_ws else None, ws.numel() if use_ws else 0, None)
        assert rc == 0, L.sctc_last_error()
        A = (a[:, :K] if akc else a[:, :M].t()).double()
        Bm = (b[:, :K].t() if bkc else b[:, :N]).double()
        ref = A @ Bm
        if bias is not None:
            ref = ref + bias.double()
        if relu:
            ref = torch.clamp(ref, min=0)
        got = c[:, :N].double()
        err = float((got - ref).abs().max() / max(1.0, float(ref.abs().max())))
        assert err < 2e-5, (case, M, N, K, akc, bkc, err)
        if pad_c:
            assert float((c[:, N:] - 7.0).abs().max()) == 0.0, "wrote beyond N"
        worst = max(worst, err)
    print("all %d GEMM cases agree (worst scaled error %.1e)" % (n_cases, worst))


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
