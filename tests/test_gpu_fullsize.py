"""The BASELINE configurations at their REAL sizes on the GPU (the workloads bench.py and
DESIGN.md quote), against the float64 oracle -- ctc_fast/debug-utils/checkgrads.py:20-40 at
production size (GPU fp32 vs CPU fp64, same weights, same data):

  cfg-3  T=1000 A=33 5x1824 TL=3 D=483 U=100, minibatch 32  -> brnn_recurrent_q_kernel<29,10>
         runs its 1000 steps; every utterance's cost and the summed gradient of every tensor
  cfg-4  T=2000 D=615 (SWBD shape), minibatch 32 (the per-GPU share of 256/8)
  cfg-5  T=8000 A=33 7x2048 TL=4 D=615 U=800, minibatch 1 and 8

Stated tolerances (fp32 device arithmetic, fp64 oracle): cost 1e-4 relative (north_star; observed
<= 2e-8).  Gradients, relative Frobenius norm per tensor, TWO comparisons:
 (a) against the oracle run with the DEVICE's ReLU / (0,maxAct) decisions imposed on its backward
     pass: 1e-4 for every tensor (observed 2e-6): this is the arithmetic-quality bar;
 (b) against the plain oracle: 3e-4, except the input-layer weight gradient dW1 = delta_1 . X^T
     at 3e-3 (observed 6e-4 at T=1000, 1.2e-3 at T=2000).  test_cfg4_input_layer_gradient_error_
     decomposed shows where (b)'s dW1 figure comes from: of 21.9 M gated units of a T=2000
     utterance TWO sit within 6e-7 layer-sigmas of their kink and land on the other side in
     float32; the delta of such a unit is present in one implementation and absent in the other.
     X is zero-mean noise, so ||dW1|| is a random-walk norm and the two stray deltas are 7e-4 of
     it, whereas the other tensors' norms carry a coherent part.  The dW1 contraction itself is
     exact to 2e-7 on the device's own operands.  (The reference's cudamat-vs-rnnetcpu check,
     debug-utils/checkgrads.py:20-40, has the same property.)
The oracle runs in forked host processes (tests/helpers.oracle_parallel)."""
import numpy as np
import pytest

from tests.helpers import oracle_parallel, oracle_variants
from tests.test_gpu_brnn import make_net, rel

TOL = {"W1": 3e-3}          # (b) plain oracle; every other tensor: TOL_DEFAULT
TOL_DEFAULT = 1.5e-4      # observed <= 9.8e-5 over rounds 2-3 (tightened from 3e-4)
TOL_MASKED = 1e-4           # (a) oracle with the device's gate decisions: every tensor


def device_masks(net, NL, TL, H, B, T):
    """i -> masks dict of utterance i for oracle.brnn.cost_and_grad, read back from the engine's
    activation matrices of the LAST call (equal-length minibatch of B utterances: packed row of
    frame t, utterance b is t*B + b)"""
    acts = {i: net.debugBuffer(i)[:, :H].reshape(T, B, H) > 0.0 for i in range(1, NL + 1) if i != TL}
    hF = net.debugBuffer(100)[:, :H].reshape(T, B, H)
    hB = net.debugBuffer(101)[:, :H].reshape(T, B, H)
    mF = (hF > 0.0) & (hF < 20.0)
    mB = (hB > 0.0) & (hB < 20.0)

    def masks_of(b):
        return {"relu": {i: a[:, b, :].T for i, a in acts.items()}, "F": mF[:, b, :].T, "B": mB[:, b, :].T}
    return masks_of


def within_tol(worst):
    return all(v < TOL.get(k, TOL_DEFAULT) for k, v in worst.items())

pytestmark = pytest.mark.gpu


def print(*a):      # observed errors also go to gpurun_out/test_notes.txt (pytest swallows stdout)
    import builtins
    import os
    builtins.print(*a)
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "test_notes.txt"), "a") as f:
            builtins.print(*a, file=f)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from nnets import brnnet
    from oracle import brnn as obrnn
    return brnnet, obrnn, torch


def tensors(net, NL):
    out = {}
    for i in range(NL + 1):
        out["W%d" % (i + 1)] = net.grad[i][0].copy_to_host().astype(np.float64)
        out["b%d" % (i + 1)] = net.grad[i][1].copy_to_host().astype(np.float64).reshape(-1)
    out["Wf"] = net.grad[NL + 1][0].copy_to_host().astype(np.float64)
    out["Wb"] = net.grad[NL + 2][0].copy_to_host().astype(np.float64)
    return out


def oracle_tensors(g, NL):
    out = {}
    for i in range(NL + 1):
        out["W%d" % (i + 1)] = g["W"][i]
        out["b%d" % (i + 1)] = np.asarray(g["b"][i]).reshape(-1)
    out["Wf"], out["Wb"] = g["Wf"], g["Wb"]
    return out


def test_cfg3_minibatch32_vs_oracle(mods):
    """the headline workload itself"""
    brnnet, obrnn, torch = mods
    D, A, H, NL, TL, T, U, B = 483, 33, 1824, 5, 3, 1000, 100, 32
    rs = np.random.RandomState(3)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    datas = [rs.randn(D, T).astype(np.float32) for _ in range(B)]
    labs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    net = make_net(brnnet, (D, A, H, NL, TL, T), params, maxUtts=B, maxBatch=T)
    costs, _, skips = net.costAndGradBatch(datas, labs)
    got = tensors(net, NL)
    assert not skips.any()
    masks_of = device_masks(net, NL, TL, H, B, T)
    c_ref, g_ref, s_ref = oracle_parallel(params, datas, labs, TL)
    assert not s_ref.any()
    np.testing.assert_allclose(costs, c_ref, rtol=1e-4)
    print("cfg3 B=32 worst cost rel err %.2e" % np.max(np.abs(costs - c_ref) / c_ref))
    want = oracle_tensors(g_ref, NL)
    worst = {k: rel(got[k], want[k]) for k in want}
    print("cfg3 B=32 gradient rel-norm errors:", {k: "%.1e" % v for k, v in worst.items()})
    assert within_tol(worst), worst
    # (a) the arithmetic-quality bar: the oracle's backward pass with the device's gate decisions
    _, g_m, _ = oracle_parallel(params, datas, labs, TL, masks_of=masks_of)
    del masks_of
    want_m = oracle_tensors(g_m, NL)
    worst_m = {k: rel(got[k], want_m[k]) for k in want_m}
    print("cfg3 B=32 gradient rel-norm errors, oracle with the device's gates:",
          {k: "%.1e" % v for k, v in worst_m.items()})
    assert all(v < TOL_MASKED for v in worst_m.values()), worst_m
    # the same workload with the contractions on the bfloat16 matrix cores (three-term split, fp32-
    # accurate: NNet(..., gemm="bf16x3")) -- same oracle, same tolerances
    net3 = make_net(brnnet, (D, A, H, NL, TL, T), params, maxUtts=B, maxBatch=T, gemm="bf16x3")
    costs3, _, skips3 = net3.costAndGradBatch(datas, labs)
    assert not skips3.any()
    np.testing.assert_allclose(costs3, c_ref, rtol=1e-4)
    got3 = tensors(net3, NL)
    worst3 = {k: rel(got3[k], want[k]) for k in want}
    print("cfg3 B=32 bf16x3: worst cost rel err %.2e; gradient rel-norm errors:" % np.max(np.abs(costs3 - c_ref) / c_ref),
          {k: "%.1e" % v for k, v in worst3.items()})
    assert within_tol(worst3), worst3
    del net3
    # run-to-run reproducibility at full size (fixed summation order, no arrival-order effects)
    net.costAndGradBatch(datas, labs)
    again = tensors(net, NL)
    for k in got:
        np.testing.assert_array_equal(got[k], again[k])
    # ragged minibatch at the same size (T_b ~ U[T/2, T]): costs vs oracle for four utterances,
    # and the gradient is the sum of two half-minibatch gradients (other recurrent kernel)
    Tr = [int(t) for t in rs.randint(T // 2, T + 1, size=B)]
    dr = [d[:, :t] for d, t in zip(datas, Tr)]
    lr = [l[:max(1, t // 10)] for l, t in zip(labs, Tr)]
    costs_r, _, skips_r = net.costAndGradBatch(dr, lr)
    g_r = tensors(net, NL)
    sel = [0, 7, 19, 31]
    c_sel, _, _ = oracle_parallel(params, [dr[i] for i in sel], [lr[i] for i in sel], TL, want_grad=False)
    np.testing.assert_allclose(costs_r[sel], c_sel, rtol=1e-4)
    net.costAndGradBatch(dr[:16], lr[:16])
    net.costAndGradBatch(dr[16:], lr[16:], accumulate=True)
    g_h = tensors(net, NL)
    worst_r = {k: rel(g_h[k], g_r[k]) for k in g_r}
    print("cfg3 ragged 32 vs 16+16 accumulate:", {k: "%.1e" % v for k, v in worst_r.items()})
    assert within_tol(worst_r), worst_r


def test_cfg4_minibatch32(mods):
    """SWBD shape, the per-GPU share (32 utterances of T=2000) of BASELINE configs[3]: four
    utterances' costs and gradients against the oracle through linearity -- the minibatch
    gradient minus the gradient of the other 28 utterances (accumulate with the four removed)
    is the four utterances' gradient"""
    brnnet, obrnn, torch = mods
    D, A, H, NL, TL, T, U, B = 615, 33, 1824, 5, 3, 2000, 200, 32
    rs = np.random.RandomState(4)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    datas = [rs.randn(D, T).astype(np.float32) for _ in range(B)]
    labs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    net = make_net(brnnet, (D, A, H, NL, TL, T), params, maxUtts=B, maxBatch=T)
    costs, _, skips = net.costAndGradBatch(datas, labs)
    g32 = tensors(net, NL)
    assert not skips.any()
    sel = [1, 12, 22, 30]
    rest = [i for i in range(B) if i not in sel]
    c_ref, g_ref, s_ref = oracle_parallel(params, [datas[i] for i in sel], [labs[i] for i in sel], TL)
    np.testing.assert_allclose(costs[sel], c_ref, rtol=1e-4)
    print("cfg4 worst cost rel err %.2e" % np.max(np.abs(costs[sel] - c_ref) / c_ref))
    costs28, _, _ = net.costAndGradBatch([datas[i] for i in rest], [labs[i] for i in rest])
    np.testing.assert_allclose(costs28, costs[rest], rtol=1e-5)      # batch-composition invariance
    g28 = tensors(net, NL)
    want = oracle_tensors(g_ref, NL)
    worst = {k: rel(g32[k] - g28[k], want[k]) for k in want}
    print("cfg4 (32 - 28) vs oracle(4):", {k: "%.1e" % v for k, v in worst.items()})
    # a difference of two fp32 sums over 64000 frames, measured against the 4-utterance part
    assert all(v < 2 * TOL.get(k, TOL_DEFAULT) for k, v in worst.items()), worst
    # the four alone, directly
    net.costAndGradBatch([datas[i] for i in sel], [labs[i] for i in sel])
    g4 = tensors(net, NL)
    worst4 = {k: rel(g4[k], want[k]) for k in want}
    print("cfg4 B=4 vs oracle:", {k: "%.1e" % v for k, v in worst4.items()})
    assert within_tol(worst4), worst4


def test_cfg5_long_utterances(mods):
    """T=8000, 7x2048, U=800 (BASELINE configs[4]) in fp32: minibatch 1 (the reference's mode)
    and 8; one utterance's cost against the oracle's forward pass + CTC, the minibatch-8 costs
    against the single-utterance ones, gradient additivity over utterances"""
    brnnet, obrnn, torch = mods
    D, A, H, NL, TL, T, U = 615, 33, 2048, 7, 4, 8000, 800
    rs = np.random.RandomState(5)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    B = 8
    datas = [rs.randn(D, T).astype(np.float32) for _ in range(B)]
    labs = [rs.randint(1, A, size=U).astype(np.int32) for _ in range(B)]
    net = make_net(brnnet, (D, A, H, NL, TL, T), params, maxUtts=B, maxBatch=T)
    c1, _, s1 = net.costAndGradBatch([datas[2]], [labs[2]])
    g1 = tensors(net, NL)
    assert not s1.any()
    # the reference's own parity check (debug-utils/checkgrads.py:20-40) compares GRADIENTS device vs
    # CPU: one T=8000 utterance, every tensor, against the plain oracle and against the oracle whose
    # backward pass uses the device's gate decisions (both variants in forked host processes)
    masks = device_masks(net, NL, TL, H, 1, T)(0)
    (c_ref, g_ref, s_ref), (_, g_msk, _) = oracle_variants(params, datas[2], labs[2], TL,
                                                           [{"masks": None}, {"masks": masks}])
    del masks
    assert not s_ref
    print("cfg5 B=1 cost %.4f oracle %.4f rel %.2e" % (c1[0], c_ref, abs(c1[0] - c_ref) / c_ref))
    assert c1[0] == pytest.approx(c_ref, rel=1e-4)
    want = oracle_tensors(g_ref, NL)
    worst = {k: rel(g1[k], want[k]) for k in want}
    print("cfg5 B=1 gradient rel-norm errors vs the plain oracle:", {k: "%.1e" % v for k, v in worst.items()})
    want_m = oracle_tensors(g_msk, NL)
    worst_m = {k: rel(g1[k], want_m[k]) for k in want_m}
    print("cfg5 B=1 gradient rel-norm errors, oracle with the device's gates:",
          {k: "%.1e" % v for k, v in worst_m.items()})
    assert all(v < TOL_MASKED for v in worst_m.values()), worst_m
    # plain oracle: tied gate decisions (float32 vs float64) put whole deltas on one side only; with
    # 131 M gated units per utterance a handful of ties is expected -- 3e-4 here (dW1 3e-3), see the
    # module docstring and test_cfg4_input_layer_gradient_error_decomposed
    assert all(v < TOL.get(k, 3e-4) for k, v in worst.items()), worst
    del g_ref, g_msk, want, want_m
    c8, _, s8 = net.costAndGradBatch(datas, labs)
    g8 = tensors(net, NL)
    assert not s8.any()
    assert c8[2] == pytest.approx(c1[0], rel=1e-5)
    # additivity: minibatch 8 == accumulate(utterance 2 alone, the other seven)
    rest = [i for i in range(B) if i != 2]
    net.costAndGradBatch([datas[2]], [labs[2]])
    c7, _, _ = net.costAndGradBatch([datas[i] for i in rest], [labs[i] for i in rest], accumulate=True)
    np.testing.assert_allclose(c7, c8[rest], rtol=1e-5)
    g17 = tensors(net, NL)
    worst = {k: rel(g17[k], g8[k]) for k in g8}
    print("cfg5 8 vs 1+7 accumulate:", {k: "%.1e" % v for k, v in worst.items()})
    assert within_tol(worst), worst
    assert all(np.isfinite(v).all() for v in g1.values())


def test_cfg1_full_size(mods):
    """BASELINE configs[0] at its real size (until round 5 only its H <= 64 golden twin ran):
    one synthetic utterance T=200, 28 symbols, 512 units, 2 layers with the temporal layer first, inputDim 615,
    minibatch 1 -- cost and every gradient tensor against the plain float64 oracle
    (ctc_fast/debug-utils/checkgrads.py:20-40 at this size)."""
    brnnet, obrnn, torch = mods
    D, A, H, NL, TL, T, U = 615, 28, 512, 2, 1, 200, 20
    rs = np.random.RandomState(1)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    data = rs.randn(D, T).astype(np.float32)
    labels = rs.randint(1, A, size=U).astype(np.int32)
    net = make_net(brnnet, (D, A, H, NL, TL, T), params)
    cost, _, skip = net.costAndGrad(data, labels)
    assert not skip
    got = tensors(net, NL)
    with np.errstate(all="ignore"):
        c_ref, g_ref, s_ref, probs_ref = obrnn.cost_and_grad(params, data.astype(np.float64), labels, TL, 20.0)
    assert not s_ref
    assert cost == pytest.approx(c_ref, rel=1e-4)
    want = oracle_tensors(g_ref, NL)
    worst = {k: rel(got[k], want[k]) for k in want}
    print("cfg1 full size: cost %.6f oracle %.6f rel err %.2e; gradient rel-norm errors:" % (cost, c_ref, abs(cost - c_ref) / c_ref),
          {k: "%.1e" % v for k, v in worst.items()})
    assert within_tol(worst), worst
    net.costAndGrad(data, labels)                         # bit-reproducible
    again = tensors(net, NL)
    for k in got:
        np.testing.assert_array_equal(got[k], again[k])
    netf = make_net(brnnet, (D, A, H, NL, TL, T), params, train=False)
    p = netf.costAndGrad(data)
    assert p.shape == (A, T) and np.abs(p - probs_ref).max() < 1e-5


@pytest.mark.parametrize("gemm", ["f32", "bf16x3"])
def test_cfg2_timit_shape_full_size(mods, gemm):
    """BASELINE configs[1] at its real size: T=300, A=62, 3x1024, temporalLayer 2, inputDim 943
    (timit-utils/runTimit.sh:21), U=30, minibatch 1 -- the reference's own mode.  Exercises, at
    size and together, what only twins covered before: brnn_recurrent_s_kernel<32,4>, an alphabet
    of 62 (one column tile, 64 padded) and split-K in the forward GEMMs (300 rows)."""
    brnnet, obrnn, torch = mods
    D, A, H, NL, TL, T, U = 943, 62, 1024, 3, 2, 300, 30
    rs = np.random.RandomState(2)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    data = rs.randn(D, T).astype(np.float32)
    labels = rs.randint(1, A, size=U).astype(np.int32)
    net = make_net(brnnet, (D, A, H, NL, TL, T), params, gemm=gemm)
    cost, _, skip = net.costAndGrad(data, labels)
    assert not skip and net.recurrentPath() == (1, 1, 0)
    got = tensors(net, NL)
    with np.errstate(all="ignore"):
        c_ref, g_ref, s_ref, probs_ref = obrnn.cost_and_grad(params, data.astype(np.float64), labels, TL, 20.0)
    assert not s_ref
    assert cost == pytest.approx(c_ref, rel=1e-4)
    want = oracle_tensors(g_ref, NL)
    worst = {k: rel(got[k], want[k]) for k in want}
    print("cfg2 full size gemm=%s: cost rel err %.2e; gradient rel-norm errors:" % (gemm, abs(cost - c_ref) / c_ref),
          {k: "%.1e" % v for k, v in worst.items()})
    assert within_tol(worst), worst
    # bit-reproducible; the forward-only model gives the oracle's probabilities
    net.costAndGrad(data, labels)
    again = tensors(net, NL)
    for k in got:
        np.testing.assert_array_equal(got[k], again[k])
    netf = make_net(brnnet, (D, A, H, NL, TL, T), params, train=False, gemm=gemm)
    p = netf.costAndGrad(data)
    assert p.shape == (A, T) and np.abs(p - probs_ref).max() < 1e-5


def test_cfg4_input_layer_gradient_error_decomposed(mods):
    """Why dW1 = delta_1 . X^T is the one tensor whose distance to the float64 oracle is ~1e-3
    (SWBD shape, T=2000) when every other tensor is within 1e-4, taken apart with the engine's
    own operands (sctc_brnn_debug_buffer):
      (1) the dW1 contraction ITSELF: device dW1 against the float64 product of the device's own
          delta_1 and X -- fp32 accumulation over 2000 frames, must be <= 2e-5;
      (2) ReLU / (0,maxAct) decisions: a unit whose pre-activation is a rounding error away from
          the kink lands on different sides in float32 and float64; its delta is then present on
          one side and absent on the other.  The flips are counted, every flipped unit is shown
          to sit on the kink (|pre-activation| tiny against the layer's scale), and
      (3) with the DEVICE's masks imposed on the oracle's backward pass every tensor, dW1
          included, agrees to 1e-4: what separates the two implementations is the placement of
          tied units, not arithmetic quality (the reference's cudamat-vs-rnnetcpu comparison,
          debug-utils/checkgrads.py:20-40, has the same property)."""
    brnnet, obrnn, torch = mods
    D, A, H, NL, TL, T, U = 615, 33, 1824, 5, 3, 2000, 200
    rs = np.random.RandomState(4)
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    data = rs.randn(D, T).astype(np.float32)
    labels = rs.randint(1, A, size=U).astype(np.int32)
    net = make_net(brnnet, (D, A, H, NL, TL, T), params)
    cost, _, skip = net.costAndGrad(data, labels)
    assert not skip
    got = tensors(net, NL)
    X = net.debugBuffer(0)[:, :D]                     # [T][D]
    d1 = net.debugBuffer(200)[:, :H]                  # [T][H]
    np.testing.assert_array_equal(X, data.T.astype(np.float64))
    # (1) the contraction alone
    dW1_64 = d1.T @ X
    e_gemm = rel(got["W1"], dW1_64)
    # (2) masks of the device
    acts = {i: net.debugBuffer(i)[:, :H].T for i in range(1, NL + 1) if i != TL}
    hF, hB = net.debugBuffer(100)[:, :H].T, net.debugBuffer(101)[:, :H].T
    masks = {"relu": {i: (a > 0.0).astype(np.float64) for i, a in acts.items()},
             "F": ((hF > 0.0) & (hF < 20.0)).astype(np.float64),
             "B": ((hB > 0.0) & (hB < 20.0)).astype(np.float64)}
    cache = {}
    with np.errstate(all="ignore"):
        c_ref, g_ref, s_ref, _ = obrnn.cost_and_grad(params, data.astype(np.float64), labels, TL, 20.0,
                                                     cache_out=cache)
        c_m, g_m, s_m, _ = obrnn.cost_and_grad(params, data.astype(np.float64), labels, TL, 20.0, masks=masks)
    assert not s_ref and not s_m and cost == pytest.approx(c_ref, rel=1e-4)
    flips, worst_tie = {}, 0.0
    for i, a in acts.items():
        o = cache["acts"][i] > 0.0
        f = o != (a > 0.0)
        flips["relu%d" % i] = int(f.sum())
        if f.any():
            worst_tie = max(worst_tie, float(np.abs(cache["pre"][i][f]).max() / np.abs(cache["pre"][i]).std()))
    for name, h_dev, h_or, pre in (("F", hF, cache["hF"], cache["preF"]), ("B", hB, cache["hB"], cache["preB"])):
        o = (h_or > 0.0) & (h_or < 20.0)
        f = o != (masks[name] > 0.5)
        flips["rec" + name] = int(f.sum())
        if f.any():
            tie = np.minimum(np.abs(pre[f]), np.abs(pre[f] - 20.0))
            worst_tie = max(worst_tie, float(tie.max() / np.abs(pre).std()))
    n_units = sum(a.size for a in acts.values()) + 2 * hF.size
    want, want_m = oracle_tensors(g_ref, NL), oracle_tensors(g_m, NL)
    worst = {k: rel(got[k], want[k]) for k in want}
    worst_m = {k: rel(got[k], want_m[k]) for k in want_m}
    d1_err = rel(d1.T, cache["d1"])
    print("cfg4 B=1 dW1 decomposition: GEMM alone %.1e | delta_1 vs oracle %.1e | mask flips %s of %d units "
          "(farthest flipped unit %.1e layer-sigmas from its kink) | vs oracle: W1 %.1e worst other %.1e | "
          "vs oracle with the device's masks: W1 %.1e worst other %.1e"
          % (e_gemm, d1_err, flips, n_units, worst_tie, worst["W1"], max(v for k, v in worst.items() if k != "W1"),
             worst_m["W1"], max(v for k, v in worst_m.items() if k != "W1")))
    assert e_gemm < 2e-5, e_gemm
    assert sum(flips.values()) <= 1e-4 * n_units and worst_tie < 1e-4, (flips, worst_tie)
    assert all(v < TOL_MASKED for v in worst_m.values()), worst_m
    assert within_tol(worst), worst
