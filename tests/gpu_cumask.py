"""CU-masked side stream next to the persistent recurrent grids (review item 2(iii)):
 1. where do the workgroups of a masked stream run (XCC / SE / SH / CU ids)?
 2. headline step (cfg-3, minibatch 32) alone vs with back-to-back GEMMs on a side stream confined to
    the CUs the two-chain grid leaves idle: phase times, wall time per step, recurrent path taken,
    side GEMM rate.  The step runs on a NON-blocking torch stream: hipExtStreamCreateWithCUMask only
    makes blocking streams, and those take turns with the legacy default stream (stream 0).
usage: gpu_cumask.py [T=1000] [layout=interleaved|blocked] [cus_per_xcc=3]"""
import collections
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd"), os.path.join(ROOT, "tools", "diag")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _sctc  # noqa: E402
import sctc_diag  # noqa: E402
from nnets import brnnet  # noqa: E402

PHASES = ["fwd_gemm", "fwd_rec", "ctc", "bwd_gemm", "bwd_rec", "other"]


def where(D, stream, n=2048, hold=300):
    out = (ctypes.c_int32 * (4 * n))()
    rc = D.sctc_diag_where(out, n, hold, stream)
    assert rc == 0, D.sctc_diag_last_error()
    a = np.array(list(out)).reshape(n, 4)
    cus = collections.Counter(map(tuple, a))
    per_xcc = collections.Counter(k[0] for k in cus)
    return len(cus), dict(sorted(per_xcc.items()))


def masked_stream(D, bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = D.sctc_diag_stream_cu_mask(words, 8, ctypes.byref(s))
    assert rc == 0, D.sctc_diag_last_error()
    return s


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    layout = sys.argv[2] if len(sys.argv) > 2 else "interleaved"
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    D = sctc_diag.lib()
    L = _sctc.lib()
    print("unmasked stream:", where(D, None))
    if layout == "interleaved":      # bit b -> XCC b % 8, CU slot b // 8
        bits = [8 * slot + x for slot in range(32 - per, 32) for x in range(8)]
    else:                            # bit b -> XCC b // 32, CU slot b % 32
        bits = [32 * x + slot for x in range(8) for slot in range(32 - per, 32)]
    side = masked_stream(D, bits)
    print("masked stream (%s, %d bits):" % (layout, len(bits)), where(D, side))
    rest = masked_stream(D, [b for b in range(256) if b not in set(bits)])
    print("complement:", where(D, rest))
    D.sctc_diag_stream_destroy(rest)

    B, Dm, A, H, NL, TL, U = 32, 483, 33, 1824, 5, 3, max(1, T // 10)
    rs = np.random.RandomState(1)
    feats = torch.randn(B * T, Dm, device="cuda")
    labels = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    Ts = [T] * B
    np.random.seed(0)
    net = brnnet.NNet(Dm, A, H, NL, T, temporalLayer=TL, maxUtts=B)
    net.initParams()
    cost0, _, _ = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
    g0 = net.grad.flat.clone()

    own = torch.cuda.Stream()          # torch's pool streams are non-blocking
    own.wait_stream(torch.cuda.current_stream())

    def phases(n=3):
        import time
        acc = np.zeros(6)
        arr = (ctypes.c_float * 6)()
        with torch.cuda.stream(own):
            L.sctc_brnn_set_profiling(net._h, 2)       # asynchronous phase timers: no sync inside a step
            t0 = time.perf_counter()
            for _ in range(n):
                net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
                L.sctc_brnn_phase_ms(net._h, arr)
                acc += np.array(list(arr))
            wall = (time.perf_counter() - t0) / n * 1e3
            L.sctc_brnn_set_profiling(net._h, 0)
        d = {k: float(v) for k, v in zip(PHASES, np.round(acc / n, 3))}
        d["wall_ms"] = round(wall, 3)
        return d

    print("alone:        ", phases(), net.recurrentPath())
    # side load: GEMMs of 3200 x 1824 x 1824 (21.3 GFLOP each) back to back on the masked stream
    M = 3200
    Aop = torch.randn(M, H, device="cuda")
    Wop = torch.randn(H, H, device="cuda")
    Cop = torch.empty(M, H, device="cuda")

    def side_gemms(n):
        for _ in range(n):
            _sctc.check(L.sctc_gemm_f32(Aop.data_ptr(), H, 1, Wop.data_ptr(), H, 1, Cop.data_ptr(), H, M, H, H,
                                        None, 0, None, 0, side), "gemm_f32")

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    sstream = torch.cuda.ExternalStream(side.value)
    torch.cuda.synchronize()
    ev[0].record(sstream); side_gemms(10); ev[1].record(sstream)
    torch.cuda.synchronize()
    alone_ms = ev[0].elapsed_time(ev[1]) / 10
    print("side GEMM alone on the masked CUs: %.3f ms each = %.1f TFLOP/s" % (alone_ms, 2 * M * H * H / alone_ms / 1e9))
    for n_side in (80,):
        torch.cuda.synchronize()
        ev[0].record(sstream); side_gemms(n_side); ev[1].record(sstream)
        ph = phases(3)
        path = net.recurrentPath()
        torch.cuda.synchronize()
        print("with %3d side GEMMs queued: " % n_side, ph, path,
              "side: %.3f ms each" % (ev[0].elapsed_time(ev[1]) / n_side))
    torch.cuda.synchronize()
    cost1, _, _ = net.costAndGradBatch(None, labels, feats_dev=feats, T_b=Ts)
    print("bit-identical after:", bool(torch.equal(net.grad.flat, g0) and np.array_equal(cost0, cost1)))
    print("shared_mode", int(L.sctc_shared_device()))


if __name__ == "__main__":
    main()
