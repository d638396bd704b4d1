"""sgd.SGD on the GPU against a NumPy restatement of the reference loop (ctc_fast/sgd.py:57-167)
fed with the oracle's float64 gradients: Nesterov look-ahead, momentum 0.5 for the first 10
iterations, global-norm clipping through the learning rate, skip handling, checkpoint format."""
import io
import pickle
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def flat(p):
    parts = []
    for w, b in zip(p["W"], p["b"]):
        parts += [w.ravel(), b.ravel()]
    parts += [p["Wf"].ravel(), p["Wb"].ravel()]
    return np.concatenate(parts)


def unflat(v, like):
    out = {"W": [], "b": [], "Wf": None, "Wb": None}
    o = 0
    for w, b in zip(like["W"], like["b"]):
        out["W"].append(v[o:o + w.size].reshape(w.shape)); o += w.size
        out["b"].append(v[o:o + b.size].reshape(b.shape)); o += b.size
    out["Wf"] = v[o:o + like["Wf"].size].reshape(like["Wf"].shape); o += like["Wf"].size
    out["Wb"] = v[o:o + like["Wb"].size].reshape(like["Wb"].shape)
    return out


def gflat(g):
    return flat({"W": g["W"], "b": g["b"], "Wf": g["Wf"], "Wb": g["Wb"]})


def test_sgd_matches_reference_loop():
    import torch
    assert torch.cuda.is_available()
    from nnets import brnnet
    import sgd
    from oracle import brnn as obrnn
    rs = np.random.RandomState(3)
    D, A, H, NL, TL, maxT = 12, 7, 32, 3, 2, 30
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    keys = ["u%d" % i for i in range(14)]
    data_dict, alis = {}, {}
    for i, k in enumerate(keys):
        T = int(rs.randint(8, maxT))
        data_dict[k] = (rs.randn(D, T) * 2.0).astype(np.float32)
        alis[k] = [str(v) for v in rs.randint(1, A, size=3)]
    alis["u5"] = ["2"] * 9
    data_dict["u5"] = data_dict["u5"][:, :12]          # 9 repeats need 17 frames -> skip
    data_dict["u7"] = rs.randn(D, maxT + 5).astype(np.float32)   # longer than maxBatch -> filtered
    alpha, momentum, clip = 1e-3, 0.9, 6.0              # small clip so that clipping is exercised

    net = brnnet.NNet(D, A, H, NL, maxT, temporalLayer=TL)
    st = [[w, b] for w, b in zip(params["W"], params["b"])] + [[params["Wf"], None], [params["Wb"], None]]
    net.setParams(st)
    opt = sgd.SGD(net, maxT, alpha=alpha, momentum=momentum, maxGradNorm=clip)
    random.seed(11)
    order = list(keys)
    opt.run(data_dict, alis, order)

    # ---- the reference loop on the oracle
    random.seed(11)
    ref_keys = list(keys)
    random.shuffle(ref_keys)
    assert ref_keys == order
    w = flat(params).astype(np.float64)
    v = np.zeros_like(w)
    it, costt, clipped = 0, [], 0
    mom = 0.5
    with np.errstate(all="ignore"):
        for k in ref_keys:
            it += 1                                   # sgd.py:71: counted before the filters
            if it > 10:
                mom = momentum
            x = data_dict[k]
            if x.shape[1] > maxT:
                continue
            lab = np.array(alis[k], dtype=np.int32)
            if x.shape[1] < lab.shape[0]:
                continue
            cost, g, skip, _ = obrnn.cost_and_grad(unflat(w + mom * v, params), x, lab, TL, 20.0)
            if skip:
                continue
            gv = gflat(g)
            gnorm = np.sqrt(np.sum(gv ** 2))
            alph = alpha * (clip / gnorm) if gnorm > clip else alpha
            clipped += gnorm > clip
            v = mom * v - alph * gv
            w = w + v
            costt.append(cost)
    assert clipped >= 3 and it == opt.it and it == len(keys)
    assert len(opt.costt) == len(costt)
    np.testing.assert_allclose(opt.costt, costt, rtol=2e-4)
    got = np.concatenate([np.concatenate([net.stack[i][0].copy_to_host().ravel(),
                                          net.stack[i][1].copy_to_host().ravel()])
                          for i in range(NL + 1)] +
                         [net.stack[NL + 1][0].copy_to_host().ravel(),
                          net.stack[NL + 2][0].copy_to_host().ravel()])
    assert np.linalg.norm(got - w) / np.linalg.norm(w) < 2e-5
    assert np.linalg.norm(got - flat(params)) / np.linalg.norm(w) > 1e-3    # it did move

    # ---- checkpoint: two consecutive pickles like runNNet.py:181-192
    buf = io.BytesIO()
    opt.toFile(buf)
    net.toFile(buf)
    buf.seek(0)
    it2, costt2, expcost2, vel = pickle.load(buf)
    assert it2 == opt.it and costt2 == opt.costt and len(vel) == NL + 3
    assert vel[0][0].shape == (H, D) and vel[-1][1].shape == (1, 1)
    buf.seek(0)
    net2 = brnnet.NNet(D, A, H, NL, maxT, temporalLayer=TL)
    opt2 = None
    net2._allocate()
    opt2 = sgd.SGD(net2, maxT, alpha=alpha, momentum=momentum, maxGradNorm=clip)
    opt2.fromFile(buf)
    net2.fromFile(buf)
    np.testing.assert_array_equal(opt2.velocity[1][0].copy_to_host(), opt.velocity[1][0].copy_to_host())
    np.testing.assert_array_equal(net2.stack[2][0].copy_to_host(), net.stack[2][0].copy_to_host())


def test_sgd_minibatch_mean_gradient():
    """minibatch=4: one step == reference step with the MEAN of the 4 utterances' gradients"""
    import torch
    from nnets import brnnet
    import sgd
    from oracle import brnn as obrnn
    rs = np.random.RandomState(4)
    D, A, H, NL, TL, maxT = 10, 6, 32, 2, 1, 24
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    keys = ["a", "b", "c", "d"]
    data_dict = {k: rs.randn(D, int(rs.randint(10, maxT))).astype(np.float32) for k in keys}
    alis = {k: [str(v) for v in rs.randint(1, A, size=2)] for k in keys}
    net = brnnet.NNet(D, A, H, NL, maxT, temporalLayer=TL, maxUtts=4)
    st = [[w, b] for w, b in zip(params["W"], params["b"])] + [[params["Wf"], None], [params["Wb"], None]]
    net.setParams(st)
    opt = sgd.SGD(net, maxT, alpha=1e-2, momentum=0.9, maxGradNorm=1e9, minibatch=4)
    random.seed(2)
    order = list(keys)
    opt.run(data_dict, alis, order)
    with np.errstate(all="ignore"):
        costs, g, skips, n_valid = obrnn.cost_and_grad_batch(
            params, [data_dict[k] for k in order],
            [np.array(alis[k], dtype=np.int32) for k in order], TL, mean=True)
    w = flat(params) - 1e-2 * gflat(g)
    got = np.concatenate([np.concatenate([net.stack[i][0].copy_to_host().ravel(),
                                          net.stack[i][1].copy_to_host().ravel()])
                          for i in range(NL + 1)] +
                         [net.stack[NL + 1][0].copy_to_host().ravel(),
                          net.stack[NL + 2][0].copy_to_host().ravel()])
    assert opt.it == 1
    assert np.linalg.norm(got - w) / np.linalg.norm(w) < 1e-6
    assert opt.costt[0] == pytest.approx(float(np.mean(costs)), rel=1e-4)


@pytest.mark.parametrize("mb", [1, 4])
def test_sgd_l2_term_enters_once(mb):
    """reg > 0: one step with minibatch 1 (reference semantics: reg*W inside costAndGrad,
    brnnet.py:197-198,244-247) and with minibatch 4 (reg*W added once after the 1/n_valid
    scaling) against the oracle's  mean(data gradients) + reg*W ; biases carry no L2 term; the
    bookkept cost contains the L2 cost in both modes (brnnet.py:178-183)"""
    from nnets import brnnet
    import sgd
    from oracle import brnn as obrnn
    rs = np.random.RandomState(14)
    D, A, H, NL, TL, maxT = 10, 6, 32, 2, 1, 24
    reg = 0.05
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    for b in params["b"]:
        b += rs.randn(*b.shape)             # non-zero biases: an L2 term on them would show
    keys = ["a", "b", "c", "d"][:mb]
    data_dict = {k: rs.randn(D, int(rs.randint(10, maxT))).astype(np.float32) for k in keys}
    alis = {k: [str(v) for v in rs.randint(1, A, size=2)] for k in keys}
    net = brnnet.NNet(D, A, H, NL, maxT, temporalLayer=TL, maxUtts=mb, reg=reg)
    st = [[w, b] for w, b in zip(params["W"], params["b"])] + [[params["Wf"], None], [params["Wb"], None]]
    net.setParams(st)
    opt = sgd.SGD(net, maxT, alpha=1e-2, momentum=0.9, maxGradNorm=1e9, minibatch=mb)
    random.seed(2)
    order = list(keys)
    opt.run(data_dict, alis, order)
    with np.errstate(all="ignore"):
        costs, g, skips, n_valid = obrnn.cost_and_grad_batch(
            params, [data_dict[k] for k in order],
            [np.array(alis[k], dtype=np.int32) for k in order], TL, reg=reg, mean=True)
    assert n_valid == mb
    w = flat(params) - 1e-2 * gflat(g)
    got = np.concatenate([np.concatenate([net.stack[i][0].copy_to_host().ravel(),
                                          net.stack[i][1].copy_to_host().ravel()])
                          for i in range(NL + 1)] +
                         [net.stack[NL + 1][0].copy_to_host().ravel(),
                          net.stack[NL + 2][0].copy_to_host().ravel()])
    assert np.linalg.norm(got - w) / np.linalg.norm(w) < 1e-6
    mats = list(params["W"]) + [params["Wf"], params["Wb"]]
    regcost = sum(0.5 * reg * float(np.sum(m * m)) for m in mats)
    assert opt.costt[0] == pytest.approx(float(np.mean(costs)) + regcost, rel=1e-4)
    assert opt.regcost[0] == pytest.approx(regcost, rel=1e-5)
    gn = np.linalg.norm(gflat(g))
    assert opt.last_gnorm == pytest.approx(gn, rel=1e-4)


@pytest.mark.parametrize("mb", [1, 4])
def test_sgd_l2_term_at_the_look_ahead_point(mb):
    """reg > 0 over THREE steps (velocity != 0 from the second step on): the reference evaluates the
    L2 term reg*W at the Nesterov look-ahead point w + mom*v (sgd.py:91-95 moves the weights there
    before costAndGrad, brnnet.py:197-198 adds reg*W); the minibatch path, which adds the term after
    the 1/n_valid scaling with the look-ahead already undone, must use the same point"""
    from nnets import brnnet
    import sgd
    from oracle import brnn as obrnn
    rs = np.random.RandomState(15)
    D, A, H, NL, TL, maxT = 10, 6, 32, 2, 1, 24
    reg, alpha = 5.0, 2e-3                  # a large L2 term: a wrong evaluation point must show
                                            # (a larger step makes fp32 softmax underflow -> skips)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    for b in params["b"]:
        b += rs.randn(*b.shape)
    keys = ["k%d" % i for i in range(3 * mb)]
    data_dict = {k: rs.randn(D, int(rs.randint(10, maxT))).astype(np.float32) for k in keys}
    alis = {k: [str(v) for v in rs.randint(1, A, size=2)] for k in keys}
    net = brnnet.NNet(D, A, H, NL, maxT, temporalLayer=TL, maxUtts=mb, reg=reg)
    st = [[w, b] for w, b in zip(params["W"], params["b"])] + [[params["Wf"], None], [params["Wb"], None]]
    net.setParams(st)
    opt = sgd.SGD(net, maxT, alpha=alpha, momentum=0.9, maxGradNorm=1e9, minibatch=mb)
    random.seed(5)
    order = list(keys)
    opt.run(data_dict, alis, order)
    assert opt.it == 3
    w = flat(params).astype(np.float64)
    v = np.zeros_like(w)
    w_plain = w.copy()                      # the same loop with reg*W taken at w (the old behaviour)
    v_plain = v.copy()
    mom = 0.5
    with np.errstate(all="ignore"):
        for i in range(0, len(order), mb):
            ks = order[i:i + mb]
            xs = [data_dict[k] for k in ks]
            ls = [np.array(alis[k], dtype=np.int32) for k in ks]
            _, g, _, n_valid = obrnn.cost_and_grad_batch(unflat(w + mom * v, params), xs, ls, TL, reg=reg, mean=True)
            assert n_valid == mb
            v = mom * v - alpha * gflat(g)
            w = w + v
            _, g0, _, _ = obrnn.cost_and_grad_batch(unflat(w_plain + mom * v_plain, params), xs, ls, TL, reg=0.0, mean=True)
            _, gr, _, _ = obrnn.cost_and_grad_batch(unflat(w_plain, params), xs, ls, TL, reg=reg, mean=True)
            _, gn, _, _ = obrnn.cost_and_grad_batch(unflat(w_plain, params), xs, ls, TL, reg=0.0, mean=True)
            v_plain = mom * v_plain - alpha * (gflat(g0) + gflat(gr) - gflat(gn))
            w_plain = w_plain + v_plain
    got = np.concatenate([np.concatenate([net.stack[i][0].copy_to_host().ravel(),
                                          net.stack[i][1].copy_to_host().ravel()])
                          for i in range(NL + 1)] +
                         [net.stack[NL + 1][0].copy_to_host().ravel(),
                          net.stack[NL + 2][0].copy_to_host().ravel()])
    err = np.linalg.norm(got - w) / np.linalg.norm(w)
    wrong = np.linalg.norm(w_plain - w) / np.linalg.norm(w)
    assert wrong > 20 * max(err, 1e-7), (err, wrong)      # the test can tell the two apart
    assert err < 5e-6, err


def test_reference_py2_checkpoint_loads(golden):
    """a params.pk in the reference's own byte format (Python-2 cPickle protocol 0,
    sgd.py:36-42 + brnnet.py:258-267; tests/golden/ref_py2_params.pk) resumes: SGD state and
    velocity, then the network weights"""
    import os
    from nnets import brnnet
    import sgd
    from tests.conftest import GOLDEN
    g = golden("ref_py2_params.npz")
    D, A, H, NL, TL = [int(v) for v in g["dims"]]
    net = brnnet.NNet(D, A, H, NL, 16, temporalLayer=TL)
    net.initParams()
    opt = sgd.SGD(net, 16)
    with open(os.path.join(GOLDEN, "ref_py2_params.pk"), "rb") as f:
        opt.fromFile(f)
        net.fromFile(f)
    assert opt.it == int(g["it"])
    np.testing.assert_array_equal(opt.costt, g["costt"])
    np.testing.assert_array_equal(opt.expcost, g["expcost"])
    for i in range(NL + 3):
        np.testing.assert_array_equal(net.stack[i][0].copy_to_host(), g["w%d" % i])
        np.testing.assert_array_equal(opt.velocity[i][0].copy_to_host(), g["vw%d" % i])
        if i <= NL:
            np.testing.assert_array_equal(net.stack[i][1].copy_to_host(), g["b%d" % i])
            np.testing.assert_array_equal(opt.velocity[i][1].copy_to_host(), g["vb%d" % i])
    # and a forward-only model built from the same file (runNNet.py:215-219 test mode)
    net2 = brnnet.NNet(D, A, H, NL, 16, train=False, temporalLayer=TL)
    with open(os.path.join(GOLDEN, "ref_py2_params.pk"), "rb") as f:
        import pickle as _p
        _p.load(f, encoding="latin1")
        net2.fromFile(f)
    np.testing.assert_array_equal(net2.stack[0][0].copy_to_host(), g["w0"])
