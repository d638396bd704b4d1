#!/usr/bin/env python3
"""GPU diagnostics (not a test): device info, kernel micro-timings, phase breakdown of
the cfg-3 step.  Writes to stdout; run on the GPU box:  python tests/gpu_diag.py [sections]"""
import ctypes
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import _sctc  # noqa: E402
from tools.diag import sctc_diag  # noqa: E402

PHASES = ["fwd_gemm", "fwd_rec", "ctc", "bwd_gemm", "bwd_rec", "other"]


def timed(fn, iters=5, warm=2):
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


def sec_info():
    L = _sctc.lib()
    cu, lds, mem = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    name = ctypes.create_string_buffer(128)
    L.sctc_device_info(ctypes.byref(cu), ctypes.byref(lds), ctypes.byref(mem), name, 128)
    print("device:", name.value.decode(), "CUs", cu.value, "LDS/CU", lds.value, "mem GB",
          mem.value / 2 ** 30)
    print("host cores:", os.cpu_count())
    print("selftest mask:", sctc_diag.lib().sctc_selftest(None))


def sec_gemm():
    L = _sctc.lib()
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for (M, N, K, akc, bkc, tag) in ((32000, 1824, 1824, 1, 1, "fwd NT"),
                                     (32000, 1824, 1824, 1, 0, "dgrad NN"),
                                     (1824, 1824, 32000, 0, 0, "wgrad TN"),
                                     (32000, 1824, 512, 1, 1, "fwd in-layer"),
                                     (64, 1824, 32000, 0, 0, "wgrad out-layer (split-K)"),
                                     (8192, 8192, 8192, 1, 1, "square 8k NT")):
        a = torch.randn((M, K) if akc else (K, M), device="cuda")
        b = torch.randn((N, K) if bkc else (K, N), device="cuda")
        c = torch.empty((M, N), device="cuda")

        def run():
            rc = L.sctc_gemm_f32(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc,
                                 c.data_ptr(), N, M, N, K, None, 0, ws.data_ptr(), ws.numel(), None)
            assert rc == 0, L.sctc_last_error()
        ms = timed(run)
        print("gemm %-28s M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s" %
              (tag, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
        del a, b, c


def sec_gemm2():
    """shape sweep around the cfg-3 forward GEMM"""
    L = _sctc.lib()
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for (M, N, K) in ((32000, 1824, 1824), (32000, 1792, 1824), (32000, 1920, 1824), (32768, 2048, 2048),
                      (8192, 1824, 1824), (32000, 1824, 8192), (8192, 8192, 1824), (3072, 1824, 1824),
                      (6144, 1824, 1824), (12288, 1824, 1824)):
        a = torch.randn((M, K), device="cuda")
        b = torch.randn((N, K), device="cuda")
        c = torch.empty((M, N), device="cuda")

        def run():
            rc = L.sctc_gemm_f32(a.data_ptr(), K, 1, b.data_ptr(), K, 1, c.data_ptr(), N, M, N, K, None, 0,
                                 ws.data_ptr(), ws.numel(), None)
            assert rc == 0
        ms = timed(run)
        print("gemm NT M=%d N=%d K=%d blocks=%d: %.3f ms  %.1f TFLOP/s" %
              (M, N, K, ((M + 127) // 128) * ((N + 127) // 128), ms, 2.0 * M * N * K / ms / 1e9))
        del a, b, c


def sec_gemm3():
    """does the N=1824 slowdown come from the partial tile or from the C row stride?"""
    L = _sctc.lib()
    M, K = 32000, 1824
    for (N, ldc, lda) in ((1824, 1824, 1824), (1824, 1920, 1824), (1920, 1920, 1824), (1824, 2048, 1824),
                          (1824, 1824, 1856), (1824, 1856, 1856), (1792, 1792, 1824), (1824, 1888, 1824)):
        a = torch.randn((M, lda), device="cuda")
        b = torch.randn((N, lda), device="cuda")
        c = torch.empty((M, ldc), device="cuda")

        def run():
            rc = L.sctc_gemm_f32(a.data_ptr(), lda, 1, b.data_ptr(), lda, 1, c.data_ptr(), ldc, M, N, K, None, 0,
                                 None, 0, None)
            assert rc == 0
        ms = timed(run)
        print("gemm NT M=%d N=%d K=%d ldc=%d lda=ldb=%d: %.3f ms  %.1f TFLOP/s" %
              (M, N, K, ldc, lda, ms, 2.0 * M * N * K / ms / 1e9))
        del a, b, c


def sec_gemmk():
    """time vs K at fixed M x N: the intercept is the per-block fixed cost (prologue+epilogue)"""
    L = _sctc.lib()
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for (M, N) in ((32768, 1536), (32000, 1824), (32000, 1792)):
        for K in (64, 256, 512, 1024, 2048, 4096, 8192):
            a = torch.randn((M, K), device="cuda")
            b = torch.randn((N, K), device="cuda")
            c = torch.empty((M, N), device="cuda")

            def run():
                rc = L.sctc_gemm_f32(a.data_ptr(), K, 1, b.data_ptr(), K, 1,
                                     c.data_ptr(), N, M, N, K, None, 0, ws.data_ptr(), ws.numel(), None)
                assert rc == 0, L.sctc_last_error()
            ms = timed(run)
            print("gemmk M=%d N=%d K=%5d: %.3f ms  %.1f TFLOP/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
            del a, b, c


def sec_twostream():
    """experiment: two half-minibatches (2 x 16 utterances) on two streams/threads vs one
    minibatch of 32 -- does GEMM work of one half fill the matrix pipes while the other half
    sits in its latency-bound recurrence?"""
    import threading
    from nnets import brnnet
    D, A, H, NL, TL, T, U = 483, 33, 1824, 5, 3, 1000, 100
    rs = np.random.RandomState(1)

    def make(B):
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
        net.initParams()
        feats = torch.randn(B * T, D, device="cuda")
        labels = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
        return net, feats, labels, [T] * B

    steps = 6
    one = make(32)
    for _ in range(2):
        one[0].costAndGradBatch(None, one[2], feats_dev=one[1], T_b=one[3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one[0].costAndGradBatch(None, one[2], feats_dev=one[1], T_b=one[3])
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    print("one stream  B=32: %.2f ms/step  %.0f frames/s" % (t1 / steps * 1e3, 32 * T * steps / t1))
    del one
    halves = [make(16), make(16)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def worker(i, n):
        net, feats, labels, Ts = halves[i]
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
            streams[i].synchronize()

    for n in (2, steps):
        th = [threading.Thread(target=worker, args=(i, n)) for i in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        t2 = time.perf_counter() - t0
    print("two streams 2xB=16: %.2f ms per pair of half-steps  %.0f frames/s" %
          (t2 / steps * 1e3, 32 * T * steps / t2))
    # one half alone, for reference
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    worker(0, steps)
    t3 = time.perf_counter() - t0
    print("one stream  B=16: %.2f ms/step  %.0f frames/s" % (t3 / steps * 1e3, 16 * T * steps / t3))


def sec_fabric():
    """hand-off latencies between workgroups (same XCD / different XCDs)"""
    L = sctc_diag.lib()
    for rep in range(2):
        out = (ctypes.c_float * 10)()
        rc = L.sctc_probe_fabric(out, 10, None)
        assert rc == 0, L.sctc_diag_last_error()
        v = list(out)
        print("partners: same-XCD block %d, cross-XCD block %d" % (v[0], v[1]))
        print("  flag ping-pong us/round trip  same XCD: sc0 %.2f  sc1 %.2f  sc0+sc1 %.2f" % tuple(v[2:5]))
        print("  flag ping-pong us/round trip cross XCD: sc0 %.2f  sc1 %.2f  sc0+sc1 %.2f" % tuple(v[5:8]))
        print("  tagged 1 KiB payload us/round trip: same %.2f  cross %.2f" % tuple(v[8:10]))


def sec_handoff():
    """is flag-after-drain + immediate plain loads safe? (tools/diag/handoff_probe.hip)"""
    L = sctc_diag.lib()
    for rep in range(3):
        out = (ctypes.c_float * 9)()
        assert L.sctc_probe_handoff(out, 9, None) == 0, L.sctc_diag_last_error()
        v = list(out)
        for m, name in enumerate(("drain + plain loads (shipped protocol)", "NO drain + plain loads", "drain + sc1 loads")):
            print("handoff %-40s stale dwords %8d of %d   iterations %d   %.2f us/iteration"
                  % (name, v[3 * m], 15 * 20000 * 256, v[3 * m + 1], v[3 * m + 2]))


def sec_mfmarate():
    """sustained fp32 MFMA rate without memory traffic (what the clock under load allows)"""
    L = sctc_diag.lib()
    out = (ctypes.c_float * 8)()
    assert L.sctc_probe_mfma(out, 8, None) == 0, L.sctc_diag_last_error()
    print("pure MFMA loops, 3 waves/SIMD on all CUs, constant operands: 32x32x2 f32 %.1f TFLOP/s (%.2f ms), "
          "16x16x4 f32 %.1f TFLOP/s (%.2f ms); nominal peak 157.3 at 2.4 GHz" % tuple(out[:4]))
    print("   random operands per MFMA group:                         32x32x2 f32 %.1f TFLOP/s (%.2f ms), "
          "16x16x4 f32 %.1f TFLOP/s (%.2f ms)" % tuple(out[4:]))


def sec_sgdloop():
    """end-to-end SGD loop (sgd.SGD.run: host features -> H2D -> costAndGrad -> fused Nesterov step)
    against the bare costAndGrad step times"""
    import logging
    import sgd
    from nnets import brnnet
    logging.getLogger().setLevel(logging.WARNING)
    D, A, H, NL, TL, T, U = 483, 33, 1824, 5, 3, 1000, 100
    rs = np.random.RandomState(3)
    for (mb, n_utts) in ((32, 128), (1, 24)):
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T + 1, temporalLayer=TL, maxUtts=max(1, mb))
        net.initParams()
        opt = sgd.SGD(net, T + 1, alpha=1e-5, momentum=0.95, minibatch=mb)
        keys = ["u%03d" % i for i in range(n_utts)]
        import dataLoader
        shard = dataLoader._host_buffer(n_utts * T, D)      # what DataLoader.loadDataFileDict returns
        shard[:] = rs.randn(n_utts * T, D)
        data = {k: shard[i * T:(i + 1) * T].T for i, k in enumerate(keys)}
        alis = {k: rs.randint(1, A, size=U).astype(np.int32) for k in keys}
        warm = keys[:max(mb, 4)]
        opt.run(data, alis, list(warm))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.run(data, alis, list(keys))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = n_utts // max(1, mb)
        print("sgd loop minibatch=%d: %d utterances in %.1f ms -> %.0f frames/s, %.2f ms per step" %
              (mb, n_utts, dt * 1e3, n_utts * T / dt, dt / steps * 1e3))
        del opt, net


def sec_gemmstamp():
    """per-K-tile timeline of one GEMM block (needs a library built with -DSCTC_GEMM_STAMP)"""
    L = _sctc.lib()
    for (M, N, K, akc, bkc, tag) in ((32000, 1920, 1824, 1, 1, "NT"), (32000, 1920, 1824, 1, 0, "NN"),
                                     (1920, 1920, 4096, 0, 0, "TN")):
        a = torch.randn((M, K) if akc else (K, M), device="cuda")
        b = torch.randn((N, K) if bkc else (K, N), device="cuda")
        c = torch.empty((M, N), device="cuda")
        ws = torch.zeros(4096, dtype=torch.int32, device="cuda")
        # a tiny workspace: too small for split-K, so splits stays 1 and the stamps land in it
        for _ in range(3):
            rc = L.sctc_gemm_f32(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc,
                                 c.data_ptr(), N, M, N, K, None, 0, ws.data_ptr(), 0, None)
            assert rc == 0
        torch.cuda.synchronize()
        st = ws.cpu().numpy()[:16 * 8].reshape(16, 8).astype(np.int64)
        per = np.diff(st[:, 0]) & 0xffffffff
        wall = np.diff(st[:, 5]) & 0xffffffff            # 100 MHz constant clock
        mhz = 100.0 * per.sum() / max(1, wall.sum())
        print("gemm %s: shader cycles per K tile median %d (ideal 6144 = 3 waves x 32 MFMA x 64) | "
              "mfma+interleaved %d, barrier %d | shader clock under load ~%.0f MHz" %
              (tag, np.median(per), np.median((st[:, 3] - st[:, 0]) & 0xffffffff),
               np.median((st[:, 4] - st[:, 3]) & 0xffffffff), mhz))


def sec_ctc():
    import ctc_fast
    rs = np.random.RandomState(0)
    for (B, T, U, A) in ((32, 1000, 100, 33), (1, 1000, 100, 33), (256, 1000, 100, 33),
                         (1024, 1000, 100, 33), (4096, 1000, 100, 33),
                         (32, 2000, 200, 33), (8, 8000, 800, 33)):
        logits = torch.randn(B * T, A, device="cuda")
        probs = torch.softmax(logits, dim=1).contiguous()
        seqs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]

        def run():
            ctc_fast.ctc_loss_batch(probs, seqs, lengths=[T] * B)
        ms = timed(run, iters=3, warm=1)
        alg = B * (8.0 * A * T + 4 * U + 8)
        print("ctc f32 B=%d T=%d U=%d: %.3f ms (incl. host wrapper)  alg %.2f MB -> %.1f GB/s" %
              (B, T, U, ms, alg / 1e6, alg / ms / 1e6))


def sec_gemmh():
    """the 16-bit-operand GEMM (sctc_gemm_h16) at the cfg-5 shapes, against the fp32 GEMM"""
    L = _sctc.lib()
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for (M, N, K, akc, bkc, tag) in ((64000, 2048, 2048, 1, 1, "fwd NT"),
                                     (64000, 2048, 2048, 1, 0, "dgrad NN"),
                                     (2048, 2048, 64000, 0, 0, "wgrad TN"),
                                     (8000, 2048, 2048, 1, 1, "fwd NT (B=1)"),
                                     (2048, 2048, 8000, 0, 0, "wgrad TN (B=1)"),
                                     (8192, 8192, 8192, 1, 1, "square 8k NT")):
        a = torch.randn((M, K) if akc else (K, M), device="cuda")
        b = torch.randn((N, K) if bkc else (K, N), device="cuda")
        c = torch.empty((M, N), device="cuda")
        res = []
        for dt in (None, _sctc.F16, _sctc.BF16):
            def run():
                if dt is None:
                    rc = L.sctc_gemm_f32(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc,
                                         c.data_ptr(), N, M, N, K, None, 0, ws.data_ptr(), ws.numel(), None)
                else:
                    rc = L.sctc_gemm_h16(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc,
                                         c.data_ptr(), N, M, N, K, None, 0, dt, ws.data_ptr(), ws.numel(), None)
                assert rc == 0, L.sctc_last_error()
            ms = timed(run, iters=5, warm=2)
            res.append((ms, 2.0 * M * N * K / ms / 1e9))
        byts = 4.0 * (M * K + K * N + M * N)
        print("%-18s M=%d N=%d K=%d: f32 %.3f ms %.0f TF | f16 %.3f ms %.0f TF (%.2f TB/s of operand bytes) | bf16 %.3f ms %.0f TF"
              % (tag, M, N, K, res[0][0], res[0][1], res[1][0], res[1][1], byts / res[1][0] / 1e9, res[2][0], res[2][1]))


def sec_gemmx3():
    """the three-term bfloat16 split GEMM (sctc_gemm_h16, SCTC_BF16X3): speed at the cfg-3 shapes and
    error against a float64 product, next to the fp32 matrix-core GEMM"""
    L = _sctc.lib()
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")

    def call(dt, a, b, c, M, N, K, akc, bkc):
        if dt is None:
            rc = L.sctc_gemm_f32(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc,
                                 c.data_ptr(), N, M, N, K, None, 0, ws.data_ptr(), ws.numel(), None)
        else:
            rc = L.sctc_gemm_h16(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc,
                                 c.data_ptr(), N, M, N, K, None, 0, dt, ws.data_ptr(), ws.numel(), None)
        assert rc == 0, L.sctc_last_error()
    for (M, N, K, akc, bkc, tag) in ((32000, 1824, 1824, 1, 1, "fwd NT"), (32000, 1824, 1824, 1, 0, "dgrad NN"),
                                     (1824, 1824, 32000, 0, 0, "wgrad TN"), (32000, 1824, 484, 1, 1, "fwd L1"),
                                     (64000, 2048, 2048, 1, 1, "fwd NT cfg5"), (8192, 8192, 8192, 1, 1, "square 8k")):
        a = torch.randn((M, K) if akc else (K, M), device="cuda")
        b = torch.randn((N, K) if bkc else (K, N), device="cuda")
        c = torch.empty((M, N), device="cuda")
        res = []
        for dt in (None, _sctc.BF16X3):
            ms = timed(lambda: call(dt, a, b, c, M, N, K, akc, bkc), iters=5, warm=2)
            res.append((ms, 2.0 * M * N * K / ms / 1e9))
        print("%-12s M=%d N=%d K=%d: f32 %.3f ms %.0f TF | bf16x3 %.3f ms %.0f TF (x%.2f)"
              % (tag, M, N, K, res[0][0], res[0][1], res[1][0], res[1][1], res[0][0] / res[1][0]))
        del a, b, c
    # accuracy: |C - C64| relative to sum_k |a||b| (the bound both error models are stated in)
    for (M, N, K, akc, bkc, scale) in ((1024, 512, 2048, 1, 1, "randn"), (512, 1024, 4096, 0, 0, "randn"),
                                       (1024, 512, 2048, 1, 0, "lognormal")):
        a = torch.randn((M, K) if akc else (K, M), device="cuda")
        b = torch.randn((N, K) if bkc else (K, N), device="cuda")
        if scale == "lognormal":
            a = a * torch.exp(4 * torch.randn_like(a))
            b = b * torch.exp(4 * torch.randn_like(b))
        A64 = (a if akc else a.t()).double()
        B64 = (b if bkc else b.t()).double()
        ref = A64 @ B64.t()
        bound = A64.abs() @ B64.abs().t()
        out = []
        for dt in (None, _sctc.BF16X3, _sctc.BF16):
            c = torch.empty((M, N), device="cuda")
            call(dt, a, b, c, M, N, K, akc, bkc)
            e = ((c.double() - ref).abs() / bound)
            out.append((e.max().item(), e.mean().item()))
        print("error / sum|a||b| (%s, K=%d, %s%s): f32 mfma max %.2e mean %.2e | bf16x3 max %.2e mean %.2e | "
              "bf16 (one term) max %.2e mean %.2e" % (scale, K, "N" if akc else "T", "T" if bkc else "N",
                                                      out[0][0], out[0][1], out[1][0], out[1][1], out[2][0], out[2][1]))


def sec_gemmx3t():
    """timing only (kernel-variant experiments via SCTC_LIB_PATH)"""
    L = _sctc.lib()
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    out = []
    for (M, N, K, akc, bkc, tag) in ((64000, 2048, 2048, 1, 1, "fwdNT"), (2048, 2048, 64000, 0, 0, "wgradTN")):
        a = torch.randn((M, K) if akc else (K, M), device="cuda")
        b = torch.randn((N, K) if bkc else (K, N), device="cuda")
        c = torch.empty((M, N), device="cuda")

        def run():
            rc = L.sctc_gemm_h16(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc,
                                 c.data_ptr(), N, M, N, K, None, 0, _sctc.BF16X3, ws.data_ptr(), ws.numel(), None)
            assert rc == 0, L.sctc_last_error()
        ms = timed(run, iters=5, warm=2)
        out.append("%s %.3f ms %.0f TF" % (tag, ms, 2.0 * M * N * K / ms / 1e9))
    print(os.environ.get("SCTC_LIB_PATH", "default"), "|", " | ".join(out))


def sec_ragged3():
    """ragged minibatch (index-gathered recurrent weight gradient): fp32 vs bf16x3 step time"""
    from nnets import brnnet
    D, A, H, NL, TL, T, U, B = 483, 33, 1824, 5, 3, 1000, 100, 32
    rs = np.random.RandomState(7)
    Ts = sorted([int(t) for t in rs.randint(T // 2, T + 1, size=B)], reverse=True)
    feats = torch.randn(sum(Ts), D, device="cuda")
    labels = [rs.randint(1, A, size=max(1, t // 10)).astype(np.int32) for t in Ts]
    for gemm in ("f32", "bf16x3"):
        np.random.seed(0)
        net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B, gemm=gemm)
        net.initParams()
        ms = timed(lambda: net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts), iters=3, warm=1)
        print("ragged cfg3 B=32 (%d frames) %s: %.2f ms per step -> %.0f frames/s" % (sum(Ts), gemm, ms, sum(Ts) / ms * 1e3))
        L = _sctc.lib()
        L.sctc_brnn_set_profiling(net._h, 1)
        net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
        arr = (ctypes.c_float * 6)()
        L.sctc_brnn_phase_ms(net._h, arr)
        print("  phases ms:", ", ".join("%s %.2f" % (n, v) for n, v in zip(PHASES, arr)))
        del net


def sec_s3stamp():
    """cycles per K-tile step of gemm_s3_kernel and the shader clock under its load (needs a library
    built with -DSCTC_GEMM_STAMP -DSCTC_S3_STAMP)"""
    L = _sctc.lib()
    for (M, N, K, akc, bkc, tag) in ((64000, 2048, 2048, 1, 1, "NT"), (64000, 2048, 2048, 1, 0, "NN"),
                                     (2048, 2048, 16000, 0, 0, "TN")):
        a = torch.randn((M, K) if akc else (K, M), device="cuda")
        b = torch.randn((N, K) if bkc else (K, N), device="cuda")
        c = torch.empty((M, N), device="cuda")
        ws = torch.zeros(4096, dtype=torch.int32, device="cuda")
        for _ in range(3):
            rc = L.sctc_gemm_h16(a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc,
                                 c.data_ptr(), N, M, N, K, None, 0, _sctc.BF16X3, ws.data_ptr(), 0, None)
            assert rc == 0, L.sctc_last_error()
        torch.cuda.synchronize()
        st = ws.cpu().numpy()[:256].reshape(64, 4).astype(np.int64)
        st = st[st[:, 3] == 1]
        cyc = st[:, 0] / st[:, 2]
        mhz = 100.0 * st[:, 0] / np.maximum(1, st[:, 1])
        print("gemm_s3 %s: %.0f shader cycles per 24-MFMA step (768 = matrix core saturated by ONE of the two "
              "resident blocks, 1536 by both), shader clock %.0f MHz (%d blocks stamped)"
              % (tag, np.median(cyc), np.median(mhz), len(st)))


def sec_brnn(cfgname="cfg3", B=32, sync=None, fp16=False, gemm=None):
    from nnets import brnnet
    cfgs = {"cfg1": (615, 28, 512, 2, 1, 200, 20), "cfg2": (943, 62, 1024, 3, 2, 300, 30),
            "cfg3": (483, 33, 1824, 5, 3, 1000, 100), "cfg4": (615, 33, 1824, 5, 3, 2000, 200),
            "cfg5": (615, 33, 2048, 7, 4, 8000, 800)}
    D, A, H, NL, TL, T, U = cfgs[cfgname]
    if sync is not None:
        os.environ["SCTC_REC_SYNC"] = str(sync)
    np.random.seed(0)
    net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B, fp16=fp16, gemm=gemm)
    net.initParams()
    rs = np.random.RandomState(1)
    feats = torch.randn(B * T, D, device="cuda")
    labels = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    Ts = [T] * B
    L = _sctc.lib()

    def run():
        c, g, s = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
        return c, s
    t0 = time.time()
    c, s = run()
    torch.cuda.synchronize()
    print("%s%s B=%d sync=%s first call %.1f ms; cost[0]=%.4f skip=%d" %
          (cfgname, " fp16" if fp16 else (" " + gemm if gemm else ""), B, os.environ.get("SCTC_REC_SYNC", "0"), (time.time() - t0) * 1e3, c[0], s.sum()))
    ms = timed(lambda: run(), iters=3, warm=1)
    tot, gm, rc = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    mb, keep = net._minibatch(feats, Ts, labels)
    L.sctc_brnn_flops(net._h, ctypes.byref(mb), ctypes.byref(tot), ctypes.byref(gm), ctypes.byref(rc))
    print("  step %.2f ms -> %.0f frames/s, %.1f TFLOP/s overall (%.2f TFLOP: gemm %.2f rec %.2f)" %
          (ms, B * T / ms * 1e3, tot.value / ms / 1e9, tot.value / 1e12, gm.value / 1e12, rc.value / 1e12))
    L.sctc_brnn_set_profiling(net._h, 1)
    run()
    arr = (ctypes.c_float * 6)()
    L.sctc_brnn_phase_ms(net._h, arr)
    print("  phases ms:", ", ".join("%s %.2f" % (n, v) for n, v in zip(PHASES, arr)))
    L.sctc_brnn_set_profiling(net._h, 0)
    del net


def sec_recdbg(sync=0, B=32):
    """per-step timeline of the recurrent kernel (workgroups 0 and last of direction 0)"""
    from nnets import brnnet
    os.environ["SCTC_REC_DEBUG"] = "1"
    os.environ["SCTC_REC_SYNC"] = str(sync)
    D, A, H, NL, TL, T, U = 483, 33, 1824, 5, 3, 200, 20
    np.random.seed(0)
    net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
    net.initParams()
    rs = np.random.RandomState(1)
    feats = torch.randn(B * T, D, device="cuda")
    labels = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    for _ in range(2):
        net.costAndGradBatch(None, labels, feats_dev=feats, T_b=[T] * B)
    W = 2 * 16 * 8 + 512 * 8
    buf = np.zeros(2 * W, dtype=np.uint32)
    _sctc.lib().sctc_brnn_debug_read(net._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    both = buf.reshape(2, W).astype(np.int64)
    st = both[:, :256].reshape(2, 2, 16, 8)
    # every workgroup's wall-clock (10 ns) stamps of step 70: who publishes when, who sees it when
    for ps, pname in enumerate(("forward", "bptt")):
        a = both[ps, 256:].reshape(512, 8)
        a = a[a[:, 7] != 0]
        for c in sorted(set(a[:, 7])):
            w = a[a[:, 7] == c]
            c = c - 1
            t0 = w[:, 0].min()
            rel = (w[:, :6] - t0) * 0.01
            names6 = ["step start", "flags seen", "mfma done", "reduced", "stored", "published"]
            print("%s chain %d (%d wgs), us after first wg started step 70:" % (pname, c, len(w)))
            for k in range(6):
                print("    %-11s min %.2f  median %.2f  max %.2f" % (names6[k], rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
    # who is slow?  publish time of step 70 by XCD (block b runs on XCD b % 8) and by position in the grid
    for ps, pname in enumerate(("forward", "bptt")):
        a = both[ps, 256:].reshape(512, 8)
        ids = np.nonzero(a[:, 7] != 0)[0]
        w = a[ids]
        if len(w) == 0:          # a kernel without the all-workgroup stamps (the sentinel kernels)
            continue
        t0 = w[:, 0].min()
        pub = (w[:, 5] - t0) * 0.01
        seen = (w[:, 1] - t0) * 0.01
        start = (w[:, 0] - t0) * 0.01
        print("%s: step-70 publish time (us after the first start) by XCD:" % pname,
              " ".join("%d:%.1f/%.1f" % (x, np.median(pub[ids % 8 == x]), pub[ids % 8 == x].max()) for x in range(8)))
        order = np.argsort(-pub)[:12]
        print("   slowest 12 workgroups (block, xcd, chain, start, flags seen, mfma done, published):",
              "; ".join("%d,%d,%d,%.1f,%.1f,%.1f,%.1f" % (ids[o], ids[o] % 8, w[o, 7] - 1, start[o], seen[o],
                                                        (w[o, 2] - t0) * 0.01, pub[o]) for o in order))
        q = np.argsort(ids)
        dur = (w[:, 5] - w[:, 1]) * 0.01      # flags seen -> published
        print("   flags-seen -> published duration: median %.2f, p90 %.2f, max %.2f us; by XCD median:" %
              (np.median(dur), np.percentile(dur, 90), dur.max()),
              " ".join("%.2f" % np.median(dur[ids % 8 == x]) for x in range(8)))
    # placement (tools/build_variant.sh where recurrent.hip "-DSCTC_REC_WHERE"): which workgroups share a compute unit
    for ps, pname in enumerate(("forward", "bptt")):
        a = both[ps, 256:].reshape(512, 8)
        ids = np.nonzero(a[:, 7] != 0)[0]
        w = a[ids]
        if not w[:, 6].any():
            continue
        t0 = w[:, 0].min()
        pub = (w[:, 5] - t0) * 0.01
        dur = (w[:, 5] - w[:, 1]) * 0.01
        groups = {}
        for n, key in enumerate(w[:, 6]):
            groups.setdefault(int(key), []).append(n)
        sizes = np.array([len(v) for v in groups.values()])
        per_xcc = {}
        for key, v in groups.items():
            per_xcc.setdefault(key >> 16, []).append(len(v))
        print("%s placement: %d CUs hold workgroups: %s; CUs per XCC / workgroups: %s" % (
            pname, len(groups), {int(k): int((sizes == k).sum()) for k in sorted(set(sizes))},
            " ".join("%d:%d/%d" % (x, len(v), sum(v)) for x, v in sorted(per_xcc.items()))))
        alone = [v[0] for v in groups.values() if len(v) == 1]
        pairs = [v for v in groups.values() if len(v) == 2]
        same = [v for v in pairs if w[v[0], 7] == w[v[1], 7]]
        paired = [n for v in pairs for n in v]
        print("   alone on a CU: %d wgs, flags-seen->published median %.2f us, publish median %.2f / max %.2f" % (
            len(alone), np.median(dur[alone]) if alone else -1, np.median(pub[alone]) if alone else -1,
            pub[alone].max() if alone else -1))
        print("   two on a CU: %d wgs (%d pairs of the SAME chain), flags-seen->published median %.2f us, publish median %.2f / max %.2f" % (
            len(paired), len(same), np.median(dur[paired]) if paired else -1, np.median(pub[paired]) if paired else -1,
            pub[paired].max() if paired else -1))
        ph_names = ["start->flags seen", "flags seen->mfma done", "mfma done->reduced", "reduced->stored", "stored->published"]
        for nm, grp in (("alone", alone), ("two on a CU", paired)):
            if grp:
                d = np.diff(w[grp][:, :6], axis=1) * 0.01
                print("   %-12s phases (us, median): %s" % (nm, ", ".join("%s %.2f" % (n, v) for n, v in zip(ph_names, np.median(d, axis=0)))))
        blocks_alone = sorted(int(ids[n]) for n in alone)
        print("   blocks alone:", blocks_alone[:64])
    names = ["wait", "load+mfma", "lds-reduce", "epilogue", "publish"]
    if B <= 3:                   # brnn_recurrent_s_kernel stamps: start, rows staged (poll + LDS + barrier), FMAs + DPP, stored
        names = ["poll+stage", "fma+dpp", "store", "-", "-"]
    for ps, pname in enumerate(("forward", "bptt")):
        for w in range(2):
            d = np.diff(st[ps, w, :, :6], axis=1) & 0xffffffff
            step = (np.diff(st[ps, w, :, 0]) & 0xffffffff)
            print("sync=%d %s wg%s: ticks/step median %d; phases median:" % (sync, pname, "0" if w == 0 else "last", np.median(step)),
                  ", ".join("%s %d" % (n, v) for n, v in zip(names, np.median(d[1:], axis=0))))
    del net
    os.environ["SCTC_REC_DEBUG"] = "0"


def main():
    want = sys.argv[1:] or ["info", "gemm", "ctc", "brnn"]
    table = {"info": sec_info, "gemm": sec_gemm, "gemm2": sec_gemm2, "gemm3": sec_gemm3, "gemmstamp": sec_gemmstamp, "gemmk": sec_gemmk, "twostream": sec_twostream, "fabric": sec_fabric, "handoff": sec_handoff, "sgdloop": sec_sgdloop, "mfmarate": sec_mfmarate, "ctc": sec_ctc,
             "brnn": lambda: sec_brnn("cfg3", 32, 0),
             "brnn1": lambda: sec_brnn("cfg3", 32, 1),
             "recdbg": lambda: sec_recdbg(0), "recdbg1": lambda: sec_recdbg(1), "recdbgB1": lambda: sec_recdbg(1, 1),
             "recdbg64": lambda: sec_recdbg(1, 64), "recdbg128": lambda: sec_recdbg(1, 128),
             "brnn_small": lambda: sec_brnn("cfg2", 1, 0),
             "brnn4": lambda: sec_brnn("cfg4", 32, None),
             "brnn5": lambda: sec_brnn("cfg5", 1, None), "brnn5b": lambda: sec_brnn("cfg5", 8, None),
             "brnn5h": lambda: sec_brnn("cfg5", 1, None, True), "brnn5bh": lambda: sec_brnn("cfg5", 8, None, True),
             "brnn3h": lambda: sec_brnn("cfg3", 32, None, True), "gemmh": sec_gemmh, "gemmx3": sec_gemmx3, "s3stamp": sec_s3stamp, "ragged3": sec_ragged3, "gemmx3t": sec_gemmx3t,
             "brnn3x": lambda: sec_brnn("cfg3", 32, None, False, "bf16x3"), "brnn4x": lambda: sec_brnn("cfg4", 32, None, False, "bf16x3"),
             "brnn2x": lambda: sec_brnn("cfg2", 1, None, False, "bf16x3"),
             "brnn2": lambda: sec_brnn("cfg2", 1, None), "brnnB": lambda: [sec_brnn("cfg3", b, None) for b in (1, 2, 4, 8, 16, 32)], "brnn1u": lambda: sec_brnn("cfg3", 1, None)}
    for name in want:
        print("==== %s" % name, flush=True)
        try:
            table[name]()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()


if __name__ == "__main__":
    main()
