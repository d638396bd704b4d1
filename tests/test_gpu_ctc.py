"""GPU parity of the CTC kernels (through the C ABI / the ctc_fast surface) against
the CPU oracle and the golden vectors of the reference.  Run with -m gpu on an MI355X.

Tolerances: the float64 device path is compared at 1e-10 relative (same arithmetic,
different summation order of the per-frame normaliser); the float32 path at the
north_star tolerance 1e-4 relative on the loss (observed ~1e-6) and 2e-4 relative
norm on the gradient.
"""
import numpy as np
import pytest

from tests.helpers import mid_input, softmax0, time_trials_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import _sctc
    import ctc_fast
    from oracle import ctc as octc
    return _sctc, ctc_fast, octc, torch


def test_selftest_primitives(mods):
    from tools.diag import sctc_diag        # probes live outside the product library
    mask = sctc_diag.lib().sctc_selftest(None)
    assert mask == 0, "failed probes bitmask %d (1 shr, 2 sum, 4 gather, 8 mfma16, 16 mfma32)" % mask


def test_ctc_f64_tiny_golden(mods, golden):
    _, cf, octc, _ = mods
    g = golden("ctc_tiny.npz")
    for i in range(int(g["n"])):
        y, seq = np.asfortranarray(g["y%d" % i]), g["seq%d" % i]
        cost, grad, skip = cf.ctc_loss(y, seq)
        assert not skip
        assert cost == pytest.approx(float(g["cost%d" % i]), rel=1e-11, abs=1e-12), i
        np.testing.assert_allclose(grad, g["grad%d" % i], rtol=1e-9, atol=1e-12)
        assert grad.flags.f_contiguous and grad.dtype == np.float64 and grad.shape == y.shape


def test_ctc_f64_time_trials_known_answer(mods, golden):
    _, cf, octc, _ = mods
    g = golden("ctc_time_trials.npz")
    p, seq = time_trials_input()
    cost, grad, skip = cf.ctc_loss(np.asfortranarray(p), seq)
    assert not skip
    assert cost == pytest.approx(1710.233966660, abs=1e-6)
    assert cost == pytest.approx(float(g["cost"]), rel=1e-11)
    np.testing.assert_allclose(grad[:, ::37], g["grad_stride37"], rtol=1e-7, atol=1e-11)
    assert np.abs(grad).sum() == pytest.approx(float(g["sum_abs_grad"]), rel=1e-9)
    assert np.abs(grad.sum(axis=0)).max() < 1e-10


def test_ctc_f64_mid_and_long(mods, golden):
    _, cf, octc, _ = mods
    g = golden("ctc_mid.npz")
    for T, U in ((1000, 100), (2000, 200)):
        logits, seq = mid_input(T, 33, U, 0)
        y = np.asfortranarray(softmax0(logits))
        cost, grad, skip = cf.ctc_loss(y, seq)
        assert not skip
        assert cost == pytest.approx(float(g["T%d_cost" % T]), rel=1e-11)
        np.testing.assert_allclose(grad[:, ::41], g["T%d_grad_stride41" % T], rtol=1e-7, atol=1e-11)
    # cfg-5 shape (T=8000, U=800 -> 32 states per lane): the reference's own numbers (G8
    # fixture), then every element against the oracle
    logits, seq = mid_input(8000, 33, 800, 0)
    y = np.asfortranarray(softmax0(logits))
    cost, grad, skip = cf.ctc_loss(y, seq)
    assert not skip
    assert cost == pytest.approx(float(g["T8000_cost"]), rel=1e-11)
    assert cost == pytest.approx(25384.546219, abs=1e-3)
    np.testing.assert_allclose(grad[:, ::163], g["T8000_grad_stride41"], rtol=1e-6, atol=1e-10)
    c_ref, g_ref, s_ref = octc.ctc_loss(y, seq)
    assert not s_ref
    np.testing.assert_allclose(grad, g_ref, rtol=1e-6, atol=1e-10)


def test_ctc_skip_and_quirks(mods, golden):
    _, cf, octc, _ = mods
    g = golden("ctc_skip.npz")
    seq = g["rep_seq"]
    for T in (4, 5, 6, 7, 8):
        y = np.asfortranarray(g["rep_y_T%d" % T])
        cost, grad, skip = cf.ctc_loss(y, seq)
        assert skip == bool(g["rep_skip_T%d" % T]), T
        if skip:
            assert not grad.any()                 # the reference's zero-initialised grad
        else:
            assert cost == pytest.approx(float(g["rep_cost_T%d" % T]), rel=1e-11)
            np.testing.assert_allclose(grad, g["rep_grad_T%d" % T], rtol=1e-9, atol=1e-12)
    _, _, skip = cf.ctc_loss(np.asfortranarray(g["zero_y"]), g["zero_seq"])
    assert skip
    # T < U: empty band -> +inf cost, grad = params, skip False
    cost, grad, skip = cf.ctc_loss(np.asfortranarray(g["short_y"]), g["short_seq"])
    assert not skip and np.isinf(cost) and cost > 0
    np.testing.assert_allclose(grad, g["short_y"], rtol=1e-12)
    # T = 1 quirk
    t = golden("ctc_tiny.npz")
    cost, _, _ = cf.ctc_loss(np.asfortranarray(t["y9"]), t["seq9"])
    assert cost == pytest.approx(float(t["cost9"]), rel=1e-12)


def test_ctc_argument_rejection(mods):
    _, cf, _, _ = mods
    y = np.asfortranarray(softmax0(np.random.RandomState(0).randn(4, 6)))
    seq = np.array([1, 2], dtype=np.int32)
    with pytest.raises(ValueError):
        cf.ctc_loss(np.ascontiguousarray(y), seq)
    with pytest.raises(ValueError):
        cf.ctc_loss(y.astype(np.float32), seq)
    with pytest.raises(ValueError):
        cf.ctc_loss(y, seq.astype(np.int64))
    with pytest.raises(ValueError):
        cf.ctc_loss(y, np.array([1, 9], dtype=np.int32))   # label outside the alphabet
    with pytest.raises(TypeError):
        cf.ctc_loss(None, seq)
    cf.ctc_loss(y, seq)


def test_ctc_f32_batch_vs_oracle(mods):
    """float32 kernels, ragged batch, repeats and label==blank, against the float64 oracle"""
    _, cf, octc, torch = mods
    rs = np.random.RandomState(11)
    A = 33
    shapes = [(300, 30), (257, 40), (64, 31), (1000, 100), (5, 2), (129, 64), (1, 1)]
    probs, seqs = [], []
    for T, U in shapes:
        probs.append(np.asfortranarray(softmax0(rs.randn(A, T) * 2.0)))
        s = rs.randint(0, A, size=U).astype(np.int32)     # 0 == blank id allowed
        if U > 3:
            s[1] = s[2]                                    # a repeat
        seqs.append(s)
    costs, grads, skips = cf.ctc_loss_batch([p.astype(np.float32) for p in probs], seqs)
    for p, s, c, g, k in zip(probs, seqs, costs, grads, skips):
        c_ref, g_ref, k_ref = octc.ctc_loss(np.asfortranarray(p.astype(np.float32).astype(np.float64)), s)
        assert bool(k) == k_ref
        if k_ref:
            continue
        if np.isinf(c_ref):
            assert np.isinf(c)
            continue
        assert c == pytest.approx(c_ref, rel=1e-4, abs=1e-4)          # north_star tolerance
        assert abs(c - c_ref) <= 2e-5 * max(1.0, abs(c_ref))          # what fp32 actually achieves
        assert np.linalg.norm(g - g_ref) <= 2e-4 * np.linalg.norm(g_ref) + 1e-6


def test_ctc_f32_large_alphabet(mods):
    _, cf, octc, _ = mods
    rs = np.random.RandomState(2)
    for A in (62, 65, 130, 200):
        T, U = 90, 17
        p = np.asfortranarray(softmax0(rs.randn(A, T)))
        s = rs.randint(1, A, size=U).astype(np.int32)
        costs, grads, skips = cf.ctc_loss_batch([p], [s])
        c_ref, g_ref, _ = octc.ctc_loss(p, s)
        assert costs[0] == pytest.approx(c_ref, rel=1e-9)
        np.testing.assert_allclose(grads[0], g_ref, rtol=1e-7, atol=1e-11)


def test_ctc_device_tensor_batch_f32(mods):
    _, cf, octc, torch = mods
    rs = np.random.RandomState(4)
    A, Ts, Us = 28, [200, 150], [20, 12]
    ps = [softmax0(rs.randn(A, T)) for T in Ts]
    seqs = [rs.randint(1, A, size=U).astype(np.int32) for U in Us]
    dev = torch.from_numpy(np.concatenate([p.T for p in ps], axis=0).astype(np.float32)).cuda()
    cost, grad, skip = cf.ctc_loss_batch(dev, seqs, lengths=Ts)
    assert grad.shape == dev.shape and grad.is_cuda
    o = 0
    for p, s, c, T in zip(ps, seqs, cost.cpu().numpy(), Ts):
        c_ref, g_ref, _ = octc.ctc_loss(np.asfortranarray(p), s)
        assert c == pytest.approx(c_ref, rel=2e-5)
        assert np.linalg.norm(grad[o:o + T].cpu().numpy().T - g_ref) <= 2e-4 * np.linalg.norm(g_ref)
        o += T


def test_ctc_full_size_properties(mods):
    """cfg-3/cfg-4 sized batches (B=32): size-independent properties -- every gradient column
    sums to zero, cost is invariant to the batch position, reversing time and labels gives
    the same cost (alpha/beta symmetry)."""
    _, cf, octc, torch = mods
    rs = np.random.RandomState(0)
    A, T, U, B = 33, 1000, 100, 32
    logits = rs.randn(B, T, A).astype(np.float32)
    seqs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    dev = torch.softmax(torch.from_numpy(logits).cuda().double(), dim=-1).reshape(B * T, A)
    cost, grad, skip = cf.ctc_loss_batch(dev, seqs, lengths=[T] * B)
    assert not skip.any()
    assert grad.reshape(B, T, A).sum(dim=-1).abs().max().item() < 1e-9
    # same utterance at two batch positions
    dev2 = torch.cat([dev[5 * T:6 * T], dev[0:T]])
    cost2, _, _ = cf.ctc_loss_batch(dev2, [seqs[5], seqs[0]], lengths=[T, T])
    assert cost2[0].item() == cost[5].item() and cost2[1].item() == cost[0].item()
    # time/label reversal
    rev = torch.flip(dev[0:T], dims=[0]).contiguous()
    cost3, _, _ = cf.ctc_loss_batch(rev, [seqs[0][::-1].copy()], lengths=[T])
    assert cost3[0].item() == pytest.approx(cost[0].item(), rel=1e-12)
    # and one of them against the oracle
    c_ref, g_ref, _ = octc.ctc_loss(np.asfortranarray(dev[0:T].cpu().numpy().T), seqs[0])
    assert cost[0].item() == pytest.approx(c_ref, rel=1e-11)


def test_decode_best_path(mods):
    _, cf, octc, _ = mods
    rs = np.random.RandomState(8)
    y = np.asfortranarray(softmax0(rs.randn(12, 300) * 3))
    assert cf.decode_best_path(y) == octc.decode_best_path(y)
    with pytest.raises(ValueError):
        cf.decode_best_path(np.ascontiguousarray(y))


def test_softmax_rows(mods):
    _sctc, _, _, torch = mods
    rs = np.random.RandomState(1)
    for A, ld in ((33, 64), (62, 64), (100, 128)):
        x = torch.from_numpy((rs.randn(777, ld) * 5).astype(np.float32)).cuda()
        y = torch.zeros_like(x)
        rc = _sctc.lib().sctc_softmax_rows(x.data_ptr(), y.data_ptr(), 777, A, ld, None)
        assert rc == 0
        torch.cuda.synchronize()
        ref = torch.softmax(x[:, :A].double(), dim=1)
        assert (y[:, :A].double() - ref).abs().max().item() < 5e-7
        assert y[:, A:].abs().max().item() == 0.0
