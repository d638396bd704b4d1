"""Layout x shape probe of sctc_gemm_f32 (GPU): is the weight-gradient contraction slow because of its
operand layout (both operands row-contiguous, "TN") or because of its shape (short M, N, long K, split-K)?
`python tools/gemm_layout_probe.py [tag]`; the library is chosen with SCTC_LIB_PATH (tools/build_variant.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
import _sctc  # noqa: E402


def timed(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main(tag, modes=(1, 0)):
    L = _sctc.lib()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    H, R = 1824, 32000
    cases = [  # (M, N, K, a_kcontig, b_kcontig, pad, name)
        (H, H, R, 0, 0, 0, "wgrad shape, TN (shipped)"),
        (H, H, R, 1, 0, 0, "wgrad shape, NN (A transposed copy)"),
        (H, H, R, 1, 1, 0, "wgrad shape, NT (both transposed)"),
        (H, H, R, 0, 0, 32, "wgrad shape, TN, ld + 32"),
        (H, H, R // 4, 0, 0, 0, "wgrad shape / 4 in K, TN"),
        (2 * H, 2 * H, R // 4, 0, 0, 0, "2H x 2H x R/4, TN (no split-K)"),
        (R, H, H, 0, 0, 0, "fwd shape, TN"),
        (R, H, H, 1, 0, 0, "fwd shape, NN (shipped dgrad)"),
        (R, H, H, 1, 1, 0, "fwd shape, NT (shipped fwd)"),
    ]
    for sparse in modes:
        for (M, N, K, akc, bkc, pad, name) in cases:
            a = torch.randn((M, K + pad) if akc else (K, M + pad), device="cuda")
            b = torch.randn((N, K + pad) if bkc else (K, N + pad), device="cuda")
            if sparse:
                a, b = torch.relu(a), torch.relu(b)
            c = torch.empty((M, N), device="cuda")

            def run():
                rc = L.sctc_gemm_f32(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc, c.data_ptr(), N,
                                     M, N, K, None, 0, ws.data_ptr(), ws.numel(), None)
                assert rc == 0, L.sctc_last_error()
            ms = timed(run)
            print("%s %-6s %-40s %6d x %5d x %6d: %.3f ms  %.1f TFLOP/s" %
                  (tag, "sparse" if sparse else "dense", name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
            del a, b, c


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "base", (1,) if len(sys.argv) > 2 else (1, 0))
