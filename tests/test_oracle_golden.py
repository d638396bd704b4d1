"""Pins the CPU oracle (oracle/) to the reference's own outputs (tests/golden/).

Re-creates the reference's "tests" (SURVEY.md section 4): the Cython-vs-Python
known answer of ctc/time_trials.py, the finite-difference gradchecks of
ctc/ctc.py:142-169 and rnnetcpu.py:152-196, brute-force path enumeration, and the
skip / T=1 / label==blank quirks.  CPU only.
"""
import numpy as np
import pytest

from oracle import brnn as obrnn
from oracle import ctc as octc
from tests.helpers import (brute_force_ctc, fd_grad_logits, load_net, mid_input, softmax0,
                           time_trials_input)


def test_ctc_tiny_cases_match_reference(golden):
    g = golden("ctc_tiny.npz")
    for i in range(int(g["n"])):
        y, seq = np.asfortranarray(g["y%d" % i]), g["seq%d" % i]
        cost, grad, skip = octc.ctc_loss(y, seq)
        assert not skip
        assert cost == pytest.approx(float(g["cost%d" % i]), rel=1e-13, abs=1e-13)
        np.testing.assert_allclose(grad, g["grad%d" % i], rtol=1e-12, atol=1e-14)


def test_ctc_brute_force_enumeration(golden):
    g = golden("ctc_tiny.npz")
    checked = 0
    for i in range(int(g["n"])):
        y, seq = g["y%d" % i], g["seq%d" % i]
        A, T = y.shape
        if T < 2 or 0 in seq or A ** T > 400000:
            continue
        bf = brute_force_ctc(y, seq)
        cost, _, _ = octc.ctc_loss(np.asfortranarray(y), seq)
        assert cost == pytest.approx(bf, rel=1e-10)
        assert float(g["cost%d" % i]) == pytest.approx(bf, rel=1e-10)
        checked += 1
    assert checked >= 8


def test_ctc_T1_quirk(golden):
    # T=1,U=1 gives -ln(y_blank + y_label), not -ln y_label
    g = golden("ctc_tiny.npz")
    y, seq = g["y9"], g["seq9"]
    assert y.shape[1] == 1
    cost, _, _ = octc.ctc_loss(np.asfortranarray(y), seq)
    assert cost == pytest.approx(-np.log(y[0, 0] + y[seq[0], 0]), rel=1e-14)
    assert cost == pytest.approx(float(g["cost9"]), rel=1e-14)


def test_ctc_time_trials_known_answer(golden):
    g = golden("ctc_time_trials.npz")
    p, seq = time_trials_input()
    assert p.sum() == pytest.approx(float(g["params_checksum"]), rel=1e-14)
    np.testing.assert_array_equal(seq, g["seq"])
    np.testing.assert_allclose(p[:, :4], g["params_head"], rtol=1e-14)
    cost, grad, skip = octc.ctc_loss(np.asfortranarray(p), seq)
    assert not skip
    assert cost == pytest.approx(1710.233966660, abs=1e-8)
    assert cost == pytest.approx(float(g["cost"]), rel=1e-13)
    assert np.abs(grad).sum() == pytest.approx(float(g["sum_abs_grad"]), rel=1e-11)
    np.testing.assert_allclose(grad[:, ::37], g["grad_stride37"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(grad.sum(axis=1), g["grad_rowsum"], rtol=1e-9, atol=1e-11)
    assert np.abs(grad.sum(axis=0)).max() < 1e-12               # each column sums to zero


def test_ctc_mid_size(golden):
    g = golden("ctc_mid.npz")
    for T, U in ((1000, 100), (2000, 200), (8000, 800)):
        logits, seq = mid_input(T, 33, U, 0)
        k = "T%d" % T
        st = 41 if T <= 2000 else 163
        assert logits.sum() == pytest.approx(float(g[k + "_logits_checksum"]), rel=1e-13)
        np.testing.assert_array_equal(seq, g[k + "_seq"])
        cost, grad, skip = octc.ctc_loss(np.asfortranarray(softmax0(logits)), seq)
        assert not skip
        assert cost == pytest.approx(float(g[k + "_cost"]), rel=1e-13)
        np.testing.assert_allclose(grad[:, ::st], g[k + "_grad_stride41"], rtol=1e-9, atol=1e-13)
        assert np.abs(grad).sum() == pytest.approx(float(g[k + "_sum_abs_grad"]), rel=1e-10)
        # the logits entry point is the same thing with the softmax inside
        cost2, grad2, skip2, probs2 = octc.ctc_loss_logits(np.asfortranarray(logits), seq)
        assert cost2 == pytest.approx(cost, rel=1e-12)
        np.testing.assert_allclose(grad2, grad, rtol=1e-9, atol=1e-13)
    assert float(g["T1000_cost"]) == pytest.approx(3167.051290, abs=1e-5)
    assert float(g["T2000_cost"]) == pytest.approx(6302.641853, abs=1e-5)
    assert float(g["T8000_cost"]) == pytest.approx(25384.546219, abs=1e-5)


def test_ctc_skip_cases(golden):
    g = golden("ctc_skip.npz")
    seq = g["rep_seq"]
    for T in (4, 5, 6, 7, 8):
        y = np.asfortranarray(g["rep_y_T%d" % T])
        cost, grad, skip = octc.ctc_loss(y, seq)
        assert skip == bool(g["rep_skip_T%d" % T]), T
        if not skip:
            assert cost == pytest.approx(float(g["rep_cost_T%d" % T]), rel=1e-13)
            np.testing.assert_allclose(grad, g["rep_grad_T%d" % T], rtol=1e-11, atol=1e-14)
    assert bool(g["rep_skip_T6"]) and not bool(g["rep_skip_T7"])
    _, _, skip = octc.ctc_loss(np.asfortranarray(g["zero_y"]), g["zero_seq"])
    assert skip and bool(g["zero_skip"])
    # T < U: the band [start,end) is empty, nothing is divided, the reference
    # returns cost = +inf, grad = params, skip = False (sgd.py:84-88 filters
    # these utterances before the call)
    cost, grad, skip = octc.ctc_loss(np.asfortranarray(g["short_y"]), g["short_seq"])
    assert skip == bool(g["short_skip"]) and not skip
    assert np.isinf(cost) and cost > 0 and np.isinf(float(g["short_cost"]))
    np.testing.assert_allclose(grad, g["short_grad"], rtol=1e-13)
    np.testing.assert_allclose(grad, g["short_y"], rtol=1e-13)


def test_ctc_fd_gradcheck():
    # ctc/ctc.py:142-169 (central difference wrt the pre-softmax activations), smaller shape
    rs = np.random.RandomState(33)
    A, U, T = 7, 5, 14
    logits = rs.randn(A, T)
    seq = rs.randint(1, A, size=U).astype(np.int32)

    def cost_fn(lg, s):
        return octc.ctc_loss(np.asfortranarray(softmax0(lg)), s)[0]

    _, grad, skip = octc.ctc_loss(np.asfortranarray(softmax0(logits)), seq)
    assert not skip
    num = fd_grad_logits(cost_fn, logits, seq, eps=1e-5)
    assert np.linalg.norm(num - grad) / np.linalg.norm(num + grad) < 1e-8


def test_ctc_argument_rejection():
    # the Cython memoryview signature rejects these with ValueError
    y = np.asfortranarray(softmax0(np.random.RandomState(0).randn(4, 6)))
    seq = np.array([1, 2], dtype=np.int32)
    with pytest.raises(ValueError):
        octc.ctc_loss(np.ascontiguousarray(y), seq)             # C order
    with pytest.raises(ValueError):
        octc.ctc_loss(y.astype(np.float32), seq)                # float32
    with pytest.raises(ValueError):
        octc.ctc_loss(y, seq.astype(np.int64))                  # int64 labels
    octc.ctc_loss(y, seq)


def test_decode_best_path():
    # ctc_fast.pyx:154-187 incl. the hard-coded drop of ids 1, 2, 8
    A, T = 10, 12
    path = [0, 3, 3, 0, 3, 1, 4, 4, 8, 5, 0, 5]
    y = np.full((A, T), 0.01)
    for t, k in enumerate(path):
        y[k, t] = 0.9
    hyp, align = octc.decode_best_path(np.asfortranarray(y))
    assert hyp == [3, 3, 4, 5, 5]
    assert align == [2, 4, 7, 9, 11]


def test_ctc_batch_driver_matches_single():
    rs = np.random.RandomState(3)
    ps, ss = [], []
    for T, U in ((20, 3), (35, 6), (11, 2)):
        ps.append(np.asfortranarray(softmax0(rs.randn(9, T))))
        ss.append(rs.randint(1, 9, size=U).astype(np.int32))
    costs, grads, skips = octc.ctc_loss_batch(ps, ss, nthreads=2)
    for p, s, c, g, k in zip(ps, ss, costs, grads, skips):
        c1, g1, k1 = octc.ctc_loss(p, s)
        assert c == c1 and k == k1
        np.testing.assert_array_equal(g, g1)


# ---------------------------------------------------------------- BRNN oracle

def _cmp_grads(got, want, rtol, atol):
    for a, b in zip(got["W"], want["W"]):
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)
    for a, b in zip(got["b"], want["b"]):
        np.testing.assert_allclose(a.reshape(-1), np.asarray(b).reshape(-1), rtol=rtol, atol=atol)
    np.testing.assert_allclose(got["Wf"], want["Wf"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(got["Wb"], want["Wb"], rtol=rtol, atol=atol)


def test_brnn_main_known_answer(golden):
    # rnnetcpu.py:180-194 -> COST 12.023458823
    params, grads, dims, data, labels, cost = load_net(golden("brnn_main.npz"))
    D, A, H, NL, TL, T = dims
    c, g, skip, _ = obrnn.cost_and_grad(params, data, labels, TL, max_act=None)
    assert not skip
    assert cost == pytest.approx(12.023458823, abs=1e-8)
    assert c == pytest.approx(cost, rel=1e-12)
    _cmp_grads(g, grads, rtol=1e-9, atol=1e-12)
    # with the GPU model's ceiling at 20 nothing changes here (activations are small)
    c20, g20, _, _ = obrnn.cost_and_grad(params, data, labels, TL, max_act=20.0)
    assert c20 == pytest.approx(c, rel=1e-14)


def test_brnn_main_init_replay(golden):
    # the reference seeds once (33), draws data, then the weights in the order
    # W1..W_{NL+1}, Wf, Wb (rnnetcpu.py:29-30,44-46 == brnnet.py:40-41,67-70)
    params, _, dims, data, _, _ = load_net(golden("brnn_main.npz"))
    D, A, H, NL, TL, T = dims
    np.random.seed(33)
    d2 = np.random.randn(D, T)
    np.testing.assert_array_equal(d2, data)
    p2 = obrnn.init_params(D, A, H, NL, TL)
    for a, b in zip(p2["W"], params["W"]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(p2["Wf"], params["Wf"])
    np.testing.assert_array_equal(p2["Wb"], params["Wb"])


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
def test_brnn_scaled_configs(golden, name):
    params, grads, dims, data, labels, cost = load_net(golden("brnn_cfg.npz"), name + "_")
    D, A, H, NL, TL, T = dims
    c, g, skip, _ = obrnn.cost_and_grad(params, data, labels, TL, max_act=20.0)
    assert not skip
    assert c == pytest.approx(cost, rel=1e-12)
    _cmp_grads(g, grads, rtol=1e-8, atol=1e-11)


def test_brnn_fd_gradcheck_with_ceiling_and_reg():
    # rnnetcpu.py:152-168 style forward difference, here with clip(0,20) active
    # on some units and reg>0 (the parts only brnnet.py has)
    rs = np.random.RandomState(5)
    D, A, H, NL, TL, T = 6, 5, 8, 3, 2, 9
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    params["Wf"] *= 2.0
    params["b"][TL - 1] += 6.0                      # push some units through the ceiling
    data = 3.0 * rs.randn(D, T)
    labels = np.array([1, 3, 2], dtype=np.int32)
    reg = 0.1
    c0, g, skip, _ = obrnn.cost_and_grad(params, data, labels, TL, max_act=20.0, reg=reg)
    assert not skip
    fwd, cache = obrnn.forward(params, data, TL, 20.0)
    assert (cache["hF"] >= 20.0).any() and (cache["hF"] == 0.0).any()
    eps = 1e-6
    for key, gkey in (("W", "W"), ("Wf", "Wf"), ("Wb", "Wb")):
        mats = params[key] if key == "W" else [params[key]]
        gm = g[gkey] if key == "W" else [g[gkey]]
        for m, dm in zip(mats, gm):
            for (i, j) in [(0, 0), (m.shape[0] - 1, m.shape[1] - 1), (m.shape[0] // 2, 1)]:
                old = m[i, j]
                m[i, j] = old + eps
                cp = obrnn.cost_and_grad(params, data, labels, TL, 20.0, reg)[0]
                m[i, j] = old - eps
                cm = obrnn.cost_and_grad(params, data, labels, TL, 20.0, reg)[0]
                m[i, j] = old
                assert dm[i, j] == pytest.approx((cp - cm) / (2 * eps), rel=2e-5, abs=2e-6)
