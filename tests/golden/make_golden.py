#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference, read-only).  Nothing of
the reference is copied into the repository: the unmodified Cython source
``ctc_fast/ctc-loss/ctc_fast.pyx`` is compiled in a scratch directory OUTSIDE the
repo (SURVEY.md Appendix C recipe: cythonize with language_level=2), and the
reference's NumPy CPU twin ``ctc_fast/debug-utils/rnnetcpu.py`` is converted to
Python 3 in that same scratch directory with ``expand -t 8`` + ``lib2to3``.  Only
the numeric inputs/outputs are written here as ``.npz`` files.

Usage:  python tests/golden/make_golden.py [--scratch /tmp/sctc_ref]

Fixtures:
  ctc_tiny.npz        G1/G5  small cases (incl. repeats, T==U-feasible, T=2, T=1 quirk)
  ctc_time_trials.npz G2     ctc/time_trials.py input (seed 33, A=40,U=125,T=1200)
  ctc_skip.npz        G4     infeasible / zero-probability cases (skip=True) + feasible twins
  ctc_mid.npz         G8     T=1000/2000 A=33 randn->softmax (cost + strided grad slice)
  brnn_main.npz       G3     rnnetcpu.py __main__ (seed 33; D=20,H=30,NL=3,TL=2,A=6,T=10)
  brnn_cfg.npz        G6     scaled-down cfg-1/2/3/5 shaped nets through rnnetcpu
"""
import argparse
import contextlib
import io
import os
import subprocess
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def build_reference(scratch):
    os.makedirs(scratch, exist_ok=True)
    pyx = os.path.join(REF, "ctc_fast/ctc-loss/ctc_fast.pyx")
    setup = os.path.join(scratch, "setup.py")
    with open(setup, "w") as f:
        f.write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "import numpy as np\n"
            "setup(ext_modules=cythonize([Extension('ctc_fast', [%r],\n"
            "      include_dirs=[np.get_include()])], language_level=2, build_dir=%r))\n"
            % (pyx, os.path.join(scratch, "cy")))
    if not any(n.startswith("ctc_fast.") and n.endswith(".so") for n in os.listdir(scratch)):
        subprocess.check_call([sys.executable, setup, "build_ext", "--build-lib", scratch,
                               "--build-temp", os.path.join(scratch, "tmp")], cwd=scratch,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    twin = os.path.join(scratch, "rnnetcpu_e.py")
    if not os.path.exists(twin):
        src = os.path.join(REF, "ctc_fast/debug-utils/rnnetcpu.py")
        with open(twin, "w") as f:
            subprocess.check_call(["expand", "-t", "8", src], stdout=f)
        subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", twin],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, scratch)
    import ctc_fast  # noqa: the reference's Cython module (sets np.seterr raise at import)
    import rnnetcpu_e
    return ctc_fast, rnnetcpu_e


def ref_ctc(ctc_fast, params, seq, blank=0):
    """Calls the reference; maps its Python-3 skip path (AttributeError on
    `e.message`, ctc_fast.pyx:148) to skip=True with no cost/grad."""
    p = np.asfortranarray(params, dtype=np.float64)
    s = np.ascontiguousarray(seq, dtype=np.int32)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            cost, grad, skip = ctc_fast.ctc_loss(p, s, blank=blank)
        return float(cost), np.array(grad), bool(skip)
    except AttributeError:
        return float("nan"), None, True


def softmax0(x):
    e = np.exp(x - x.max(axis=0, keepdims=True))
    return e / e.sum(axis=0, keepdims=True)


def gen_ctc_tiny(cf):
    rs = np.random.RandomState(7)
    cases = [  # (A, T, seq)
        (3, 2, [1]), (3, 3, [1, 2]), (4, 5, [1, 1, 2]), (4, 7, [2, 2, 2]), (5, 6, [1, 2, 3]),
        (5, 5, [4, 1, 4]), (3, 4, [1, 2, 1]), (4, 7, [3, 3, 1]), (5, 3, [2, 4, 1]),
        (3, 1, [1]),                      # T=1 quirk: -ln(y_blank + y_label)
        (6, 10, [0, 1, 2]),               # label == blank id, as in rnnetcpu.py:190
        (4, 6, [0, 0]),
    ]
    out = {"n": np.int64(len(cases))}
    for i, (A, T, seq) in enumerate(cases):
        y = softmax0(rs.randn(A, T))
        cost, grad, skip = ref_ctc(cf, y, seq)
        assert not skip
        out["y%d" % i] = y
        out["seq%d" % i] = np.array(seq, dtype=np.int32)
        out["cost%d" % i] = np.float64(cost)
        out["grad%d" % i] = grad
    np.savez(os.path.join(HERE, "ctc_tiny.npz"), **out)


def time_trials_input():
    """ctc/time_trials.py:13-25 restated (seed 33): peaked random distribution."""
    np.random.seed(33)
    A, U, T = 40, 125, 1200
    seq = np.floor(np.random.rand(U) * A).astype(np.int32)
    p = np.random.randn(A, T)
    p[seq, np.arange(U)] = 3
    p[0, U:] = 3
    p = np.exp(p)
    p = p / np.sum(p, axis=0)
    return p, seq


def gen_time_trials(cf):
    p, seq = time_trials_input()
    cost, grad, skip = ref_ctc(cf, p, seq)
    assert not skip and abs(cost - 1710.233966660) < 1e-6, cost
    np.savez(os.path.join(HERE, "ctc_time_trials.npz"),
             cost=np.float64(cost), sum_abs_grad=np.float64(np.abs(grad).sum()),
             grad_stride37=grad[:, ::37], grad_rowsum=grad.sum(axis=1),
             grad_colsum_max=np.float64(np.abs(grad.sum(axis=0)).max()),
             params_checksum=np.float64(p.sum()), params_head=p[:, :4], seq=seq)


def gen_skip(cf):
    rs = np.random.RandomState(1)
    out = {}
    seq = np.array([1, 1, 1, 1], dtype=np.int32)      # needs T >= 7 (3 forced blanks)
    for T in (4, 5, 6, 7, 8):
        y = softmax0(rs.randn(4, T))
        cost, grad, skip = ref_ctc(cf, y, seq)
        out["rep_y_T%d" % T] = y
        out["rep_skip_T%d" % T] = np.bool_(skip)
        if not skip:
            out["rep_cost_T%d" % T] = np.float64(cost)
            out["rep_grad_T%d" % T] = grad
    out["rep_seq"] = seq
    # a label whose probability is exactly zero in every frame
    y = softmax0(rs.randn(5, 9))
    y[3, :] = 0.0
    y /= y.sum(axis=0, keepdims=True)
    seqz = np.array([1, 3, 2], dtype=np.int32)
    cost, grad, skip = ref_ctc(cf, y, seqz)
    assert skip
    out["zero_y"] = y
    out["zero_seq"] = seqz
    out["zero_skip"] = np.bool_(skip)
    # T < U (filtered by sgd.py:84-88 before the call, but the function itself skips)
    y = softmax0(rs.randn(4, 2))
    seqs = np.array([1, 2, 3], dtype=np.int32)
    cost, grad, skip = ref_ctc(cf, y, seqs)
    out["short_y"] = y
    out["short_seq"] = seqs
    out["short_skip"] = np.bool_(skip)          # False: empty band -> log(0) -> cost = +inf
    out["short_cost"] = np.float64(cost)
    out["short_grad"] = grad
    np.savez(os.path.join(HERE, "ctc_skip.npz"), **out)


def mid_input(T, A, U, seed):
    rs = np.random.RandomState(seed)
    logits = rs.randn(A, T)
    seq = rs.randint(1, A, size=U).astype(np.int32)
    return logits, seq


def gen_mid(cf):
    out = {}
    for T, U in ((1000, 100), (2000, 200)):
        logits, seq = mid_input(T, 33, U, 0)
        y = softmax0(logits)
        cost, grad, skip = ref_ctc(cf, y, seq)
        assert not skip
        k = "T%d" % T
        out[k + "_cost"] = np.float64(cost)
        out[k + "_sum_abs_grad"] = np.float64(np.abs(grad).sum())
        out[k + "_grad_stride41"] = grad[:, ::41]
        out[k + "_grad_rowsum"] = grad.sum(axis=1)
        out[k + "_logits_checksum"] = np.float64(logits.sum())
        out[k + "_seq"] = seq
    np.savez(os.path.join(HERE, "ctc_mid.npz"), **out)


def run_twin(tw, D, A, H, NL, TL, T, data, labels, seed):
    np.random.seed(seed)
    net = tw.RNNet(D, A, H, NL, T, temporalLayer=TL)
    net.initParams()
    weights = [[np.array(w), np.array(b)] for w, b in net.stack]
    with contextlib.redirect_stdout(io.StringIO()):
        cost, grad, skip = net.costAndGrad(data, labels)
    assert not skip
    grads = [[np.array(dw), np.array(db)] for dw, db in grad]
    return float(cost), weights, grads, net


def pack_net(prefix, out, cost, weights, grads, data, labels, dims):
    out[prefix + "cost"] = np.float64(cost)
    out[prefix + "data"] = data
    out[prefix + "labels"] = labels
    out[prefix + "dims"] = np.array(dims, dtype=np.int64)   # D, A, H, NL, TL, T
    out[prefix + "n"] = np.int64(len(weights))
    for i, ((w, b), (dw, db)) in enumerate(zip(weights, grads)):
        out[prefix + "W%d" % i] = w
        out[prefix + "b%d" % i] = b
        out[prefix + "dW%d" % i] = dw
        out[prefix + "db%d" % i] = np.asarray(db).reshape(-1, 1) if np.size(db) > 1 else np.zeros((1, 1))


def gen_brnn_main(tw):
    # rnnetcpu.py:180-194
    np.random.seed(33)
    D, A, H, NL, TL, T = 20, 6, 30, 3, 2, 10
    data = np.random.randn(D, T)
    labels = np.arange(3).astype(np.int32)
    # the reference seeds once, then draws data, then the weights: replay that order
    net = tw.RNNet(D, A, H, NL, T, temporalLayer=TL)
    net.initParams()
    weights = [[np.array(w), np.array(b)] for w, b in net.stack]
    with contextlib.redirect_stdout(io.StringIO()):
        cost, grad, skip = net.costAndGrad(data, labels)
    assert not skip and abs(cost - 12.023458823) < 1e-8, cost
    grads = [[np.array(dw), np.array(db)] for dw, db in grad]
    out = {}
    pack_net("", out, cost, weights, grads, data, labels, (D, A, H, NL, TL, T))
    np.savez(os.path.join(HERE, "brnn_main.npz"), **out)


def gen_brnn_cfg(tw):
    """Scaled-down twins of cfg-1/2/3/5 (same NL/TL/A, H<=32, T<=48).  Inputs are
    scaled so every recurrent activation stays < 20: there the no-ceiling twin
    and the clipped GPU model coincide (asserted)."""
    out = {}
    shapes = {"cfg1": (16, 28, 24, 2, 1, 40, 6), "cfg2": (20, 62, 32, 3, 2, 48, 7),
              "cfg3": (21, 33, 32, 5, 3, 40, 5), "cfg5": (16, 33, 24, 7, 4, 32, 4)}
    for seed, (name, (D, A, H, NL, TL, T, U)) in enumerate(sorted(shapes.items())):
        rs = np.random.RandomState(100 + seed)
        data = rs.randn(D, T)
        labels = rs.randint(1, A, size=U).astype(np.int32)
        cost, weights, grads, net = run_twin(tw, D, A, H, NL, TL, T, data, labels, 200 + seed)
        # recompute the recurrent activations to check the ceiling is not hit
        h = data
        for i, (w, b) in enumerate(weights[:NL + 1], start=1):
            z = w @ h + b
            if i == TL:
                Wf, Wb = weights[-2][0], weights[-1][0]
                hF = np.zeros_like(z)
                hB = np.zeros_like(z)
                hF[:, 0] = np.maximum(z[:, 0], 0)
                hB[:, -1] = np.maximum(z[:, -1], 0)
                for t in range(1, T):
                    hF[:, t] = np.maximum(z[:, t] + Wf @ hF[:, t - 1], 0)
                    hB[:, T - 1 - t] = np.maximum(z[:, T - 1 - t] + Wb @ hB[:, T - t], 0)
                assert max(hF.max(), hB.max()) < 20.0, (name, hF.max(), hB.max())
                h = hF + hB
            elif i <= NL:
                h = np.maximum(z, 0)
        pack_net(name + "_", out, cost, weights, grads, data, labels, (D, A, H, NL, TL, T))
    np.savez(os.path.join(HERE, "brnn_cfg.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/sctc_ref")
    a = ap.parse_args()
    assert not os.path.abspath(a.scratch).startswith("/root/repo"), "scratch must be outside the repo"
    cf, tw = build_reference(a.scratch)
    with np.errstate(all="ignore"):
        gen_ctc_tiny(cf)
        gen_time_trials(cf)
        gen_skip(cf)
        gen_mid(cf)
        gen_brnn_main(tw)
        gen_brnn_cfg(tw)
    for n in sorted(os.listdir(HERE)):
        if n.endswith(".npz"):
            print("%-24s %7.1f KB" % (n, os.path.getsize(os.path.join(HERE, n)) / 1024.0))


if __name__ == "__main__":
    main()
