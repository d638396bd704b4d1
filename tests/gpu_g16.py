"""The LDS-DMA staged 16-bit-operand GEMM (csrc/gemm_g16.hip) through the public entry
(sctc_gemm_h16 with SCTC_OPERANDS_16BIT): correctness against a float64 product of the 16-bit
operands on ragged shapes (row / column / k tails, split-K, both layouts, both types), then speed at the
cfg-5 shapes against round 2's register-staged kernel (SCTC_G16=0 in a child process).
usage: gpu_g16.py [check] [speed]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _sctc  # noqa: E402

IN16 = 0x100


def call(L, a16, b16, c, M, N, K, kc, dt, ws, bias=None, relu=0):
    rc = L.sctc_gemm_h16(a16.data_ptr(), a16.stride(0), kc, b16.data_ptr(), b16.stride(0), kc,
                         c.data_ptr(), c.stride(0), M, N, K, None if bias is None else bias.data_ptr(), relu,
                         dt | IN16, ws.data_ptr(), ws.numel(), None)
    assert rc == 0, L.sctc_last_error()


def check():
    os.environ["SCTC_H16_TILE"] = "1"            # the 256 x 256 kernels also for small outputs
    L = _sctc.lib()
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    worst = 0.0
    cases = [(256, 256, 64), (256, 256, 128), (512, 256, 192), (300, 260, 72), (1000, 520, 200), (264, 1824, 1824),
             (2048, 2048, 4096), (1824, 1824, 8000), (40, 24, 8), (8, 8, 640), (256, 64, 2048), (777 * 8, 33 * 8, 1824),
             # the input-layer weight gradient of cfg-5 at its real shape (dW1 = delta_1^T X: 2048 x 640 -- inputDim 615 padded like the engine pads it -- over 8000 /
             # 64000 frames; VERDICT r04 weak #3: the network-level bound on dW1 is wide by nature, the contraction
             # itself is pinned here) and the output layer's (33 symbols padded to 64)
             (2048, 640, 8000), (2048, 640, 64000), (64, 2048, 64000)]
    for (M, N, K) in cases:
        for kc in (1, 0):
            for dt, tt in ((_sctc.F16, torch.float16), (_sctc.BF16, torch.bfloat16)):
                # leading dimensions: padded like the engine's matrices (multiples of 64 elements)
                pad = lambda v: (v + 63) // 64 * 64
                if kc:
                    A = torch.randn(M, pad(K), device="cuda", generator=g).to(tt)
                    B = torch.randn(N, pad(K), device="cuda", generator=g).to(tt)
                    ref = A[:, :K].double() @ B[:, :K].double().T
                else:
                    A = torch.randn(K, pad(M), device="cuda", generator=g).to(tt)
                    B = torch.randn(K, pad(N), device="cuda", generator=g).to(tt)
                    ref = A[:, :M].double().T @ B[:, :N].double()
                # poison the padding: the kernel must not let it into the sums
                if kc:
                    A[:, K:] = float("nan"); B[:, K:] = float("nan")
                else:
                    A[:, M:] = float("nan"); B[:, N:] = float("nan")
                C = torch.full((M, pad(N)), -7.0, device="cuda")
                bias = torch.randn(N, device="cuda", generator=g)
                call(L, A, B, C, M, N, K, kc, dt, ws, bias=bias)
                got = C[:, :N].double()
                want = ref + bias.double()[None, :]
                scale = (A[:, :K].double().abs() @ B[:, :K].double().abs().T) if kc else \
                        (A[:, :M].double().abs().T @ B[:, :N].double().abs())
                err = float(((got - want).abs() / (scale + 1e-30)).max())
                untouched = bool((C[:, N:] == -7.0).all())
                worst = max(worst, err)
                flag = "" if (err < 2e-6 and untouched) else "   <-- FAIL"
                print("M=%5d N=%5d K=%5d %s %s: max |C - C64| / sum|a||b| = %.2e, padding untouched %s%s"
                      % (M, N, K, "K-contig" if kc else "row-contig", "f16 " if dt == _sctc.F16 else "bf16", err, untouched, flag))
    print("worst %.2e" % worst)
    return worst < 2e-6


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def speed():
    L = _sctc.lib()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    tag = "g16" if os.environ.get("SCTC_G16", "1") != "0" else "x16"
    for (M, N, K, kc, name) in ((64000, 2048, 2048, 1, "fwd / dgrad"), (2048, 2048, 64000, 0, "wgrad"),
                                (64000, 2048, 640, 1, "input layer"), (8000, 2048, 2048, 1, "fwd B=1"),
                                (2048, 2048, 8000, 0, "wgrad B=1"), (8192, 8192, 8192, 1, "square 8k"),
                                (4096, 4096, 4096, 1, "square 4k"), (32000, 1824, 1824, 1, "cfg-3 fwd"),
                                (1824, 1824, 32000, 0, "cfg-3 wgrad")):
        for dt, tt in ((_sctc.F16, torch.float16), (_sctc.BF16, torch.bfloat16)):
            extra = int(os.environ.get("SCTC_G16_PAD", "0"))     # extra elements per row: leading dimensions off the powers of two
            pad = lambda v: (v + 63) // 64 * 64 + extra
            A = torch.randn((M, pad(K)) if kc else (K, pad(M)), device="cuda").to(tt)
            B = torch.randn((N, pad(K)) if kc else (K, pad(N)), device="cuda").to(tt)
            C = torch.empty(M, pad(N), device="cuda")
            ms = timed(lambda: call(L, A, B, C, M, N, K, kc, dt, ws))
            print("%s%s %-12s M=%5d N=%5d K=%5d %s: %.3f ms  %.0f TFLOP/s" %
                  (tag, "+pad%d" % extra if extra else "", name, M, N, K, "f16 " if dt == _sctc.F16 else "bf16", ms, 2.0 * M * N * K / ms / 1e9), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "speed"]
    ok = True
    if "check" in what:
        ok = check()
    if "speed" in what:
        speed()
        if os.environ.get("SCTC_G16", "1") != "0" and not os.environ.get("SCTC_G16_NO_CHILD"):
            env = dict(os.environ, SCTC_G16="0")
            env.pop("SCTC_H16_TILE", None)
            subprocess.call([sys.executable, os.path.abspath(__file__), "speed"], env=env)
    sys.exit(0 if ok else 1)
