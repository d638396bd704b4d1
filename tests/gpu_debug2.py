#!/usr/bin/env python3
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    sys.path.insert(0, p)
import torch
import _sctc
from tests.helpers import softmax0
np.seterr(all="ignore")

def ref_alpha(y, seq):
    A, T = y.shape; U = len(seq); L = 2*U+1
    lab = np.zeros(L, dtype=int); lab[1::2] = seq
    al = np.zeros((T, L)); cs = np.zeros(T)
    al[0,0] = y[0,0]; al[0,1] = y[seq[0],0]; c = al[0].sum(); al[0] /= c; cs[0]=c
    for t in range(1,T):
        start = max(0, L-2*(T-t))
        for s in range(start, L):
            v = al[t-1,s]
            if s>=1: v += al[t-1,s-1]
            if s%2==1 and s>=3 and lab[s]!=lab[s-2]: v += al[t-1,s-2]
            al[t,s] = v*y[lab[s],t]
        c = al[t].sum(); al[t] /= c; cs[t]=c
    return al, cs

def run(T, U, dt, seed=11):
    rs = np.random.RandomState(seed)
    A = 33
    y = softmax0(rs.randn(A, T)*2.0)
    seq = rs.randint(1, A, size=U).astype(np.int32)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    dev = torch.from_numpy(np.ascontiguousarray(y.T).astype(dt)).cuda()
    grad = torch.empty_like(dev)
    L = _sctc.lib()
    T_b = np.array([T], dtype=np.int32); U_b = np.array([U], dtype=np.int32)
    off = np.zeros(1, dtype=np.int64)
    bt = _sctc.CtcBatch(1, A, 0, _sctc.F32 if dt==np.float32 else _sctc.F64, A, _sctc.i32(T_b), _sctc.i32(U_b), _sctc.i64(off), _sctc.i32(seq), _sctc.i64(off), None)
    n = L.sctc_ctc_workspace_bytes(ctypes.byref(bt))
    ws = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cost = torch.empty(1, dtype=torch.float64, device="cuda"); skip = torch.empty(1, dtype=torch.int32, device="cuda")
    rc = L.sctc_ctc_loss_batch(ctypes.byref(bt), dev.data_ptr(), grad.data_ptr(), cost.data_ptr(), skip.data_ptr(), ws.data_ptr(), n, None)
    assert rc == 0
    torch.cuda.synchronize()
    K = 2 if 2*U+1 <= 128 else 4
    lp = 64*K
    a256 = lambda v: (v+255)//256*256
    o = a256(32) + a256(4*U) + a256(16) + a256(8)
    esz = 4 if dt==np.float32 else 8
    alpha = ws[o:o+T*lp*esz].view(tdt).view(T, lp).cpu().numpy().astype(np.float64)
    al, cs = ref_alpha(y.astype(dt).astype(np.float64), seq)
    Lh = 2*U+1
    err = np.abs(alpha[:, :Lh] - al).max(axis=1)
    bad = np.where(err > 1e-4)[0]
    print("T=%d U=%d %s cost %.4f ref %.4f first bad frame %s  max err %.3e" % (T, U, dt.__name__, cost.item(), -np.log(cs).sum(), bad[:5], err.max()))
    if len(bad):
        t = bad[0]
        s = np.argmax(np.abs(alpha[t,:Lh]-al[t]))
        print("   frame", t, "state", s, "dev", alpha[t, max(0,s-3):s+4], "ref", al[t, max(0,s-3):s+4])
        print("   sums dev", alpha[t,:lp].sum(), "beyond L:", alpha[t, Lh:].sum(), "prev frame err", err[t-1])

for T, U in ((129, 30), (200, 30), (250, 30), (257, 30), (300, 30), (300, 10), (300, 60)):
    run(T, U, np.float32)
run(300, 30, np.float64)
