"""CPU model of ctc_fused.hip's row store (no GPU needed): the meet-in-the-middle schedule keeps alpha's first
T/2 rows and beta's last T - T/2 rows and multiplies them with the other recursion's float64 registers.  This
restates that schedule in NumPy against the C oracle (oracle/ctc_ref.c = ctc_fast.pyx:13-152) with the kept rows

  float64     exact                                                 (ctc_fast.ctc_loss host signature)
  u22         bits 61..30 of the float64 pattern, round to nearest  (float32 probabilities, the BRNN path)
  float32     what "fp32 alpha/beta" would mean taken literally

and pins the two facts the kernel's design rests on: a float32 copy of a normalised row loses the gradient at
T >> 2U (the overlap of alpha and beta sits 1e-40 .. 1e-300 below the row's mass), the 32-bit format with 10
exponent bits does not; and absum must divide per STATE before it sums (ctc_fast.pyx:125-131), the per-label
form is off by 1e-3 once the products are denormal.
"""
import numpy as np

from oracle import ctc as octc


def enc22(x):
    b = np.ascontiguousarray(x).view(np.uint64)
    return ((b + np.uint64(1 << 29)) >> np.uint64(30)).astype(np.uint32)     # Store<uint32_t>::enc


def dec22(u):
    return (u.astype(np.uint64) << np.uint64(30)).view(np.float64)            # Store<uint32_t>::dec


def lattice(y, seq, blank=0):
    """normalised alpha rows [T][L] (ctc_fast.pyx:42-76), vectorised over the states of a frame"""
    A, T = y.shape
    U = len(seq)
    L = 2 * U + 1
    lab = np.full(L, blank)
    lab[1::2] = seq
    allow = np.zeros(L, bool)
    for s in range(3, L, 2):
        allow[s] = seq[(s - 1) // 2] != seq[(s - 1) // 2 - 1]
    a = np.zeros((T, L))
    a[0, 0], a[0, 1] = y[blank, 0], y[seq[0], 0]
    a[0] /= a[0, 0] + a[0, 1]
    for t in range(1, T):
        p = a[t - 1]
        n = p.copy()
        n[1:] += p[:-1]
        n[2:] += np.where(allow[2:], p[:-2], 0.0)
        n *= y[lab, t]
        n[:max(0, L - 2 * (T - t))] = 0
        a[t] = n / n.sum()
    return a, lab


def gradient(al, be, y, lab, per_state=True):
    A = y.shape[0]
    ab = al * be                                                              # :119, [T][L]
    g = np.zeros_like(y)
    for k in range(A):
        m = lab == k
        if m.any():
            g[k] = ab[:, m].sum(axis=1)                                       # :120-131
    if per_state:
        ylab = y[lab, :].T
        with np.errstate(all="ignore"):
            v = np.where(ab != 0, ab / np.where(ylab == 0, 1, ylab), 0.0)     # :125-131
        Z = v.sum(axis=1)                                                     # :133-136
    else:
        Z = np.where(g != 0, g / np.where(y == 0, 1, y), 0.0).sum(axis=0)
    tmp = y * Z
    return np.where(tmp > 0, y - g / np.where(tmp > 0, tmp, 1), y)            # :138-145


def both(y, seq):
    al, lab = lattice(y, seq)
    be_r, _ = lattice(np.asfortranarray(y[:, ::-1]), seq[::-1].copy())
    return al, be_r[::-1, ::-1], lab


def kept(al, be, store):
    Ta = al.shape[0] // 2
    al, be = al.copy(), be.copy()
    al[:Ta] = store(al[:Ta])
    be[Ta:] = store(be[Ta:])
    return al, be


def inputs(T, U, A, seed, peaked=1.0):
    rs = np.random.RandomState(seed)
    x = rs.randn(A, T) * peaked
    y = np.exp(x - x.max(0))
    y /= y.sum(0)
    y = np.asfortranarray(y.astype(np.float32).astype(np.float64))
    return y, rs.randint(1, A, size=U).astype(np.int32)


def test_roundtrip_format():
    x = np.concatenate([[0.0, 1.0, 1.0 + 2.0 ** -52, 0.5, 2.0 ** -1000, 5e-324, 1e-310],
                        np.random.RandomState(0).rand(1000) * 10.0 ** -np.random.RandomState(1).randint(0, 300, 1000)])
    back = dec22(enc22(x))
    assert back[0] == 0.0 and back[1] == 1.0 and back[3] == 0.5 and back[4] == 2.0 ** -1000
    normal = x > 1e-300
    assert np.all(np.abs(back[normal] - x[normal]) <= 2.0 ** -23 * x[normal])
    assert np.all(np.diff(dec22(enc22(np.sort(x)))) >= 0)                     # monotone


def test_float32_rows_lose_the_gradient_u22_rows_do_not():
    for T, U, seed in ((1000, 100, 1), (1000, 20, 3), (2000, 100, 4)):
        y, seq = inputs(T, U, 33, seed)
        c_ref, g_ref, skip = octc.ctc_loss(y, seq)
        assert not skip
        al, be, lab = both(y, seq)
        assert np.abs(gradient(al, be, y, lab) - g_ref).max() < 1e-12       # the model is the reference's algorithm
        a22, b22 = kept(al, be, lambda v: dec22(enc22(v)))
        a32, b32 = kept(al, be, lambda v: v.astype(np.float32).astype(np.float64))
        e22 = np.abs(gradient(a22, b22, y, lab) - g_ref).max()
        e32 = np.abs(gradient(a32, b32, y, lab) - g_ref).max()
        assert e22 < 6e-8, (T, U, e22)            # below the quantum of the float32 gradient it is written to
        assert e32 > 0.1, (T, U, e32)


def test_absum_divides_per_state():
    y, seq = inputs(1000, 100, 33, 2, peaked=4.0)
    with np.errstate(all="ignore"):
        c_ref, g_ref, skip = octc.ctc_loss(y, seq)
    al, be, lab = both(y, seq)
    assert np.abs(gradient(al, be, y, lab, per_state=True) - g_ref).max() < 1e-12
    assert np.abs(gradient(al, be, y, lab, per_state=False) - g_ref).max() > 1e-5
