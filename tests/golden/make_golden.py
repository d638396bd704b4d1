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
  brnn_cfg.npz        G6     scaled-down cfg-1/2/3/4/5 shaped nets through rnnetcpu
  loader_ref.npz + shard/  output of the reference's dataLoader.py:38-95 (lib2to3-converted in
                      scratch) on a small shard (feats1.bin / keys1.txt / alis1.txt, kept as data)
  ref_py2_params.pk + ref_py2_params.npz   a params.pk in the byte format Python-2 cPickle
                      (protocol 0, the reference's `pickle.dump(obj, fid)`) writes for
                      sgd.py:36-42 + brnnet.py:258-267, emitted opcode by opcode (no Python 2 here)
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
    ldr = os.path.join(scratch, "dataLoader_e.py")
    if not os.path.exists(ldr):
        src = os.path.join(REF, "ctc_fast/dataLoader.py")
        with open(ldr, "w") as f:
            subprocess.check_call(["expand", "-t", "8", src], stdout=f)
        subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", ldr],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        # Python-2 `int / int` is floor division (dataLoader.py:63): the one semantic fix lib2to3
        # does not make
        txt = open(ldr).read()
        assert "(self.rawsize-self.imgsize)/2" in txt
        open(ldr, "w").write(txt.replace("(self.rawsize-self.imgsize)/2",
                                         "(self.rawsize-self.imgsize)//2"))
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
    for T, U in ((1000, 100), (2000, 200), (8000, 800)):
        logits, seq = mid_input(T, 33, U, 0)
        y = softmax0(logits)
        cost, grad, skip = ref_ctc(cf, y, seq)
        assert not skip
        k = "T%d" % T
        out[k + "_cost"] = np.float64(cost)
        out[k + "_sum_abs_grad"] = np.float64(np.abs(grad).sum())
        out[k + "_grad_stride41"] = grad[:, ::(41 if T <= 2000 else 163)]
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
              "cfg3": (21, 33, 32, 5, 3, 40, 5), "cfg5": (16, 33, 24, 7, 4, 32, 4),
              "cfg4": (30, 33, 32, 5, 3, 64, 6)}
    seeds = {"cfg1": 0, "cfg2": 1, "cfg3": 2, "cfg5": 3, "cfg4": 4}   # fixed: adding a twin moves no other
    for name, (D, A, H, NL, TL, T, U) in sorted(shapes.items()):
        seed = seeds[name]
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


def gen_loader(scratch):
    """ctc_fast/dataLoader.py:38-95 on a small shard: the shard files are kept (data in the
    reference's own on-disk format: Kaldi export of util/swbd/write_feats.sh:71) together with
    what the reference's loader returns for them."""
    import dataLoader_e
    rs = np.random.RandomState(11)
    d = os.path.join(HERE, "shard")
    os.makedirs(d, exist_ok=True)
    raw, img = 15, 9
    utts = [("sw02001-A_000098-001156", 7, [3, 1, 4]), ("sw02001-B_001980-002131", 12, [1, 5]),
            ("sw02005-A_000001-000002", 1, [9]), ("sw02005-B_012345-012999", 21, [2, 2, 7, 1, 30])]
    feats = [rs.randn(T, raw).astype(np.float32) for _, T, _ in utts]
    with open(os.path.join(d, "keys1.txt"), "w") as kf, open(os.path.join(d, "alis1.txt"), "w") as af:
        for name, T, labels in utts:
            kf.write("%s %d\n" % (name, T))
            af.write("%s %s\n" % (name, " ".join(str(l) for l in labels)))
    # Kaldi's export may hold frames beyond sum(sizes) (dataLoader.py:68 slices them off)
    np.concatenate(feats + [rs.randn(3, raw).astype(np.float32)]).tofile(os.path.join(d, "feats1.bin"))
    out = {"rawsize": np.int64(raw), "imgsize": np.int64(img)}
    for tag, im in (("crop", img), ("full", raw)):
        loader = dataLoader_e.DataLoader(d + "/", raw, im)
        data_dict, alis, keys, sizes = loader.loadDataFileDict(1)
        out[tag + "_keys"] = np.array(keys)
        out[tag + "_sizes"] = np.asarray(sizes)
        for i, k in enumerate(keys):
            out["%s_data%d" % (tag, i)] = data_dict[k]
            out["%s_alis%d" % (tag, i)] = np.array(alis[k])
        mat, _, _, _ = loader.loadDataFile(1)
        out[tag + "_mat"] = np.array(mat)
    np.savez(os.path.join(HERE, "loader_ref.npz"), **out)


class Py2Pickler:
    """Emits what Python 2's cPickle writes at protocol 0 -- the reference's
    `pickle.dump(obj, fid)` with no protocol argument (sgd.py:42, brnnet.py:267) -- for the object
    shapes a params.pk holds: int, float, list, and numpy.ndarray (through
    numpy.core.multiarray._reconstruct + __setstate__ with the raw data as a Python-2 `str`:
    the S opcode that Python 3 can only read with encoding='latin1')."""

    def __init__(self):
        self.out = []
        self.memo = 0

    def put(self):
        self.out.append(b"p%d\n" % self.memo)
        self.memo += 1

    @staticmethod
    def py2_repr(b):
        q = b'"' if (b"'" in b and b'"' not in b) else b"'"
        r = bytearray(q)
        for c in b:
            ch = bytes([c])
            if ch == q or ch == b"\\":
                r += b"\\" + ch
            elif ch == b"\t":
                r += b"\\t"
            elif ch == b"\n":
                r += b"\\n"
            elif ch == b"\r":
                r += b"\\r"
            elif c < 0x20 or c >= 0x7f:
                r += b"\\x%02x" % c
            else:
                r += ch
        return bytes(r + q)

    def string(self, b):
        self.out.append(b"S" + self.py2_repr(b) + b"\n")
        self.put()

    def integer(self, v):
        self.out.append(b"I%d\n" % v)

    def save(self, obj):
        if isinstance(obj, bool):
            self.out.append(b"I01\n" if obj else b"I00\n")
        elif isinstance(obj, int):
            self.integer(obj)
        elif isinstance(obj, float):
            self.out.append(b"F" + repr(obj).encode() + b"\n")
        elif isinstance(obj, list):
            self.out.append(b"(l")
            self.put()
            if obj:
                self.out.append(b"(")
                for v in obj:
                    self.save(v)
                self.out.append(b"e")
        elif isinstance(obj, np.ndarray):
            self.ndarray(obj)
        else:
            raise TypeError(type(obj))

    def glob(self, module, name):
        self.out.append(b"c" + module + b"\n" + name + b"\n")
        self.put()

    def ndarray(self, a):
        a = np.ascontiguousarray(a)
        # numpy's ndarray.__reduce__: (_reconstruct, (ndarray, (0,), 'b'), state)
        self.glob(b"numpy.core.multiarray", b"_reconstruct")
        self.out.append(b"(")
        self.glob(b"numpy", b"ndarray")
        self.out.append(b"(")
        self.integer(0)
        self.out.append(b"t")
        self.put()
        self.string(b"b")
        self.out.append(b"t")
        self.put()
        self.out.append(b"R")
        self.put()
        # state = (version 1, shape, dtype, is_fortran, rawdata)
        self.out.append(b"(")
        self.integer(1)
        self.out.append(b"(")
        for n in a.shape:
            self.integer(int(n))
        self.out.append(b"t")
        self.put()
        self.glob(b"numpy", b"dtype")
        self.out.append(b"(")
        self.string(a.dtype.str[1:].encode())        # 'f4'
        self.integer(0)
        self.integer(1)
        self.out.append(b"t")
        self.put()
        self.out.append(b"R")
        self.put()
        self.out.append(b"(")                          # dtype.__setstate__ tuple
        self.integer(3)
        self.string(b"<")
        self.out.append(b"NNN")
        self.integer(-1)
        self.integer(-1)
        self.integer(0)
        self.out.append(b"t")
        self.put()
        self.out.append(b"b")
        self.out.append(b"I00\n")                      # C order
        self.string(a.tobytes())
        self.out.append(b"t")
        self.put()
        self.out.append(b"b")

    def dumps(self, obj):
        self.out, self.memo = [], 0
        self.save(obj)
        self.out.append(b".")
        return b"".join(self.out)


def gen_py2_pickle():
    """params.pk = SGD pickle [it, costt, expcost, velocity stack] followed by the NNet pickle
    (list of [w, b] float32 arrays; the temporal pair shares a (1,1) dummy bias) --
    runNNet.py:181-186 writes both into one file."""
    rs = np.random.RandomState(21)
    D, A, H, NL = 6, 5, 8, 2           # temporalLayer 1
    dims = [D] + [H] * NL + [A]
    def stack():
        st = [[rs.randn(m, n).astype(np.float32), rs.randn(m, 1).astype(np.float32)]
              for n, m in zip(dims[:-1], dims[1:])]
        for _ in range(2):
            st.append([rs.randn(H, H).astype(np.float32), np.zeros((1, 1), dtype=np.float32)])
        return st
    vel, par = stack(), stack()
    it, costt, expcost = 17, [float(v) for v in rs.rand(5) * 100], [float(v) for v in rs.rand(5) * 100]
    pk = Py2Pickler()
    blob = pk.dumps([it, costt, expcost, vel]) + pk.dumps(par)
    with open(os.path.join(HERE, "ref_py2_params.pk"), "wb") as f:
        f.write(blob)
    out = {"it": np.int64(it), "costt": np.array(costt), "expcost": np.array(expcost),
           "dims": np.array([D, A, H, NL, 1], dtype=np.int64)}
    for i, ((vw, vb), (w, b)) in enumerate(zip(vel, par)):
        out["vw%d" % i], out["vb%d" % i], out["w%d" % i], out["b%d" % i] = vw, vb, w, b
    np.savez(os.path.join(HERE, "ref_py2_params.npz"), **out)
    # the stream must be a valid pickle for Python 3 (latin1) and must NEED latin1
    import io, pickle
    f = io.BytesIO(blob)
    a = pickle.load(f, encoding="latin1")
    b = pickle.load(f, encoding="latin1")
    assert a[0] == it and np.array_equal(a[3][0][0], vel[0][0]) and np.array_equal(b[-1][0], par[-1][0])
    try:
        pickle.load(io.BytesIO(blob))
        raise AssertionError("a Python-2 str pickle of array data should not load as ASCII")
    except UnicodeDecodeError:
        pass


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
        gen_loader(a.scratch)
    gen_py2_pickle()
    for n in sorted(os.listdir(HERE)):
        if n.endswith(".npz"):
            print("%-24s %7.1f KB" % (n, os.path.getsize(os.path.join(HERE, n)) / 1024.0))


if __name__ == "__main__":
    main()
