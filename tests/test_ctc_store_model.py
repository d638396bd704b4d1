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


# ---- round 6: the wide fused kernel's bookkeeping (csrc/ctc_fusedw.hip), restated

def wide_block_map(i):
    """block index -> (utterance, direction), ctc_fusedw_kernel"""
    return (i >> 4) * 8 + (i & 7), (i >> 3) & 1


def test_wide_kernel_pairs_are_eight_apart_on_one_xcd():
    """the two workgroups of an utterance are blocks i and i + 8: the same XCD (block i runs on XCD i % 8) and at most 8
    apart in dispatch order, so of the resident workgroups at most 8 can be waiting for a partner that has not been
    dispatched yet (every other resident workgroup's partner is resident too, finishes, and makes room) -- the argument
    behind the bounded spin of the kernel's two meetings"""
    for B in range(1, 70):
        grid = (B + 7) // 8 * 16
        seen = {}
        for i in range(grid):
            b, d = wide_block_map(i)
            if b < B:
                assert (b, d) not in seen
                seen[(b, d)] = i
        assert len(seen) == 2 * B
        for b in range(B):
            i0, i1 = seen[(b, 0)], seen[(b, 1)]
            assert i1 - i0 == 8 and i0 % 8 == i1 % 8
        # dispatch in block order with any capacity > 8: never more than 8 resident blocks whose partner is not resident
        for cap in (9, 16, 32):
            resident = set(range(min(cap, grid)))
            waiting = [i for i in resident if wide_block_map(i)[0] < B
                       and seen[(wide_block_map(i)[0], 1 - wide_block_map(i)[1])] not in resident]
            assert len(waiting) <= 8 and len(waiting) < len(resident)


def test_wide_kernel_lds_row_layout():
    """a frame's alpha*beta row in LDS: [NLB blank states][NLB label states], state s of global lane s // K at slot
    (s // 2) of its half; the lists hold NLB + label position; the pad entry is the slot of state NST - 1, which no label row
    of the shapes the kernel takes (2U + 1 <= NST - 1) ever reaches -- its product is always +0"""
    for W, K in ((4, 2), (4, 4), (8, 2), (8, 4)):
        NT, KH = 64 * W, K // 2
        NST, NLB = NT * K, NT * K // 2
        slot = {}
        for gl in range(NT):
            for j in range(K):
                s = K * gl + j
                base = KH * gl + (j >> 1)
                slot[s] = base if j % 2 == 0 else NLB + base
        assert sorted(slot.values()) == list(range(NST))                     # a bijection onto the row
        assert all(slot[s] == s // 2 + (NLB if s % 2 else 0) for s in range(NST))
        zrow = NST - 1
        assert slot[NST - 1] == zrow
        max_L = NST - 1                                                      # the longest row the shape takes (odd)
        for U in (1, (max_L - 1) // 2):
            L = 2 * U + 1
            assert L <= max_L and NST - 1 >= L                               # state NST - 1 does not exist
            for pos in range(U):                                             # list entry of label position pos
                assert NLB + pos == slot[2 * pos + 1] != zrow
