"""The BASELINE configurations at their REAL sizes on the GPU (the workloads bench.py and
DESIGN.md quote), against the float64 oracle -- ctc_fast/debug-utils/checkgrads.py:20-40 at
production size (GPU fp32 vs CPU fp64, same weights, same data):

  cfg-3  T=1000 A=33 5x1824 TL=3 D=483 U=100, minibatch 32  -> brnn_recurrent_q_kernel<29,10>
         runs its 1000 steps; every utterance's cost and the summed gradient of every tensor
  cfg-4  T=2000 D=615 (SWBD shape), minibatch 32 (the per-GPU share of 256/8)
  cfg-5  T=8000 A=33 7x2048 TL=4 D=615 U=800, minibatch 1 and 8

Stated tolerances (fp32 device arithmetic, fp64 oracle): cost 1e-4 relative (north_star; observed
<= 2e-8); gradients, relative Frobenius norm per tensor: 3e-4 (observed <= 1e-4), except the
input-layer weight gradient dW1 = delta_1 . X^T at 3e-3 (observed 6e-4 at 32 000 frames, 1.2e-3
at 8 000..64 000 frames of T=2000 utterances): X is zero-mean noise, so the sum over all frames
of delta*x cancels to a small norm and fp32 accumulation over K = 32 000..64 000 terms shows --
the same holds for the reference's cuBLAS sgemm.  The oracle runs in forked host processes
(tests/helpers.oracle_parallel)."""
import numpy as np
import pytest

from tests.helpers import oracle_parallel
from tests.test_gpu_brnn import make_net, rel

TOL = {"W1": 3e-3}          # every other tensor: TOL_DEFAULT
TOL_DEFAULT = 3e-4


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
    c_ref, g_ref, s_ref = oracle_parallel(params, datas, labs, TL)
    assert not s_ref.any()
    np.testing.assert_allclose(costs, c_ref, rtol=1e-4)
    print("cfg3 B=32 worst cost rel err %.2e" % np.max(np.abs(costs - c_ref) / c_ref))
    want = oracle_tensors(g_ref, NL)
    worst = {k: rel(got[k], want[k]) for k in want}
    print("cfg3 B=32 gradient rel-norm errors:", {k: "%.1e" % v for k, v in worst.items()})
    assert within_tol(worst), worst
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
    c_ref, _, s_ref = oracle_parallel(params, [datas[2]], [labs[2]], TL, want_grad=False, procs=1)
    assert not s_ref.any()
    print("cfg5 B=1 cost %.4f oracle %.4f rel %.2e" % (c1[0], c_ref[0], abs(c1[0] - c_ref[0]) / c_ref[0]))
    assert c1[0] == pytest.approx(c_ref[0], rel=1e-4)
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
