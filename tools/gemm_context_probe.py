"""Why do the fp32 GEMMs of the forward / delta shape run faster inside the costAndGrad step (1.49-1.62 ms) than
in a back-to-back loop (1.77 ms)?  Candidates: operand residency (Infinity Cache warm from the producing kernel),
clock boost after a low-power phase.  `python tools/gemm_context_probe.py`"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stanford-ctc_amd"))
import torch  # noqa: E402
import _sctc  # noqa: E402

L = _sctc.lib()
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
H = 1824


def gemm(a, b, c, M, N, K, akc, bkc):
    rc = L.sctc_gemm_f32(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc, c.data_ptr(), N, M, N, K,
                         None, 0, ws.data_ptr(), ws.numel(), None)
    assert rc == 0, L.sctc_last_error()


def each(fn, pre=None, iters=8, warm=2, gap=0.0):
    """median time of single launches, each bracketed by its own events; `pre` runs (untimed) before each"""
    out = []
    for i in range(warm + iters):
        if pre:
            pre()
        if gap:
            torch.cuda.synchronize()
            time.sleep(gap)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= warm:
            out.append(e0.elapsed_time(e1))
    out.sort()
    return out[len(out) // 2]


def report(name, ms, M, N, K):
    print("%-64s %.3f ms  %.1f TFLOP/s" % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)


# 1. footprint sweep, forward shape (A [M][K] K-contiguous, B [K][N]), back to back
for M in (2000, 4000, 8000, 16000, 32000, 64000):
    a = torch.relu(torch.randn(M, H, device="cuda")); b = torch.randn(H, H, device="cuda") * 0.02
    c = torch.empty(M, H, device="cuda")
    report("NN M=%d (A %.0f MB) back to back" % (M, M * H * 4 / 1e6), each(lambda: gemm(a, b, c, M, H, H, 1, 0)), M, H, H)
    del a, c
M = 32000
a = torch.relu(torch.randn(M, H, device="cuda")); a2 = a.clone(); b = torch.randn(H, H, device="cuda") * 0.02
c = torch.empty(M, H, device="cuda"); big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
f = lambda: gemm(a, b, c, M, H, H, 1, 0)
report("NN 32000: A rewritten right before (copy_)", each(f, pre=lambda: a.copy_(a2)), M, H, H)
report("NN 32000: 1 GiB written right before (caches flushed)", each(f, pre=lambda: big.fill_(1.0)), M, H, H)
report("NN 32000: 10 ms idle before", each(f, gap=0.01), M, H, H)
report("NN 32000: 100 ms idle before", each(f, gap=0.1), M, H, H)
# chain: c of GEMM i is A of GEMM i+1 (the forward pass's pattern), timed as a whole
x = [torch.empty(M, H, device="cuda") for _ in range(3)]
x[0].copy_(a)
def chain():
    for i in range(6):
        gemm(x[i % 3], b, x[(i + 1) % 3], M, H, H, 1, 0)
report("NN 32000: chain of 6, output -> next A (per GEMM)", each(chain) / 6, M, H, H)
e = each(lambda: [f() for _ in range(50)], iters=3, warm=1) / 50
report("NN 32000: 50 back to back (per GEMM)", e, M, H, H)
# weight-gradient shape
R = 32000
at = torch.relu(torch.randn(R, H, device="cuda")); bt = torch.relu(torch.randn(R, H, device="cuda")); ct = torch.empty(H, H, device="cuda")
g = lambda: gemm(at, bt, ct, H, H, R, 0, 0)
report("TN wgrad: back to back", each(g), H, H, R)
report("TN wgrad: 10 ms idle before", each(g, gap=0.01), H, H, R)
report("TN wgrad: B rewritten right before", each(g, pre=lambda: bt.copy_(a2)), H, H, R)
