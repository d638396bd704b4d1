"""The three-term bfloat16 split of the fp32 contractions (NNet(..., gemm="bf16x3") /
sctc_brnn_config.operand_dtype = SCTC_BF16X3 / sctc_gemm_h16(..., SCTC_BF16X3)).

Every fp32 operand x is split EXACTLY into x1 + x2 + x3 (bfloat16 each, 3 x 8 significand bits);
six of the nine cross products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the dropped
x2y3 + x3y2 + x3y3 are <= 2^-23 |xy|.  The claim under test is "fp32-accurate": the SAME
tolerances as the fp32 path everywhere (cost 1e-4 -- north_star --, gradients 1e-4 relative norm
against the float64 oracle at the fixture sizes), and a GEMM error against the float64 product
that stays within a small factor of the fp32 matrix-core kernel's own error (both are printed
to gpurun_out/test_notes.txt).  This is not the 16-bit configuration of test_gpu_fp16.py: nothing
is rounded to 16 bit here."""
import numpy as np
import pytest

from tests.helpers import load_net
from tests.test_gpu_brnn import check_grads, make_net

pytestmark = pytest.mark.gpu


def print(*a):      # observed errors also go to gpurun_out/test_notes.txt
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
    import _sctc
    from nnets import brnnet
    from oracle import brnn as obrnn
    return _sctc, brnnet, obrnn, torch


def _gemm(L, _sctc, torch, dt, a, b, c, M, N, K, akc, bkc, bias, relu, ws):
    args = (a.data_ptr(), a.shape[1], akc, b.data_ptr(), b.shape[1], bkc, c.data_ptr(), c.shape[1], M, N, K,
            bias.data_ptr() if bias is not None else None, relu)
    if dt is None:
        rc = L.sctc_gemm_f32(*args, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, None)
    else:
        rc = L.sctc_gemm_h16(*args, dt, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, None)
    assert rc == 0, L.sctc_last_error()


def test_gemm_bf16x3_all_layouts(mods):
    """four operand layouts, ragged sizes, bias / relu epilogues, split-K on and off, operands over
    ~12 decades of magnitude: error against the float64 product of the UNROUNDED operands, scaled
    by sum_k |a||b| (the quantity both error models bound), next to the fp32 kernel's"""
    _sctc, _, _, torch = mods
    L = _sctc.lib()
    rs = np.random.RandomState(11)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    worst3 = worst32 = 0.0
    for (M, N, K) in ((200, 96, 64), (132, 260, 1824), (1000, 1824, 512), (64, 1824, 3000), (1824, 512, 776),
                      (32, 32, 4), (128, 128, 32), (260, 132, 36), (4, 4, 8), (516, 388, 4096)):
        for akc in (1, 0):
            for bkc in (1, 0):
                wide = bool(rs.randint(2))
                a = torch.randn((M, K + 4) if akc else (K, M + 8), device="cuda")
                b = torch.randn((N, K) if bkc else (K, N + 4), device="cuda")
                if wide:
                    a = a * torch.exp(3.0 * torch.randn_like(a))
                    b = b * torch.exp(3.0 * torch.randn_like(b))
                bias = torch.randn(N, device="cuda") if rs.rand() < 0.5 else None
                relu = int(rs.rand() < 0.5)
                use_ws = ws if rs.rand() < 0.7 else None
                A64 = (a[:, :K] if akc else a[:, :M].t()).double()
                B64 = (b[:, :K].t() if bkc else b[:, :N]).double()
                ref = A64 @ B64
                bound = A64.abs() @ B64.abs()
                if bias is not None:
                    ref = ref + bias.double()
                    bound = bound + bias.double().abs()
                if relu:
                    ref = torch.clamp(ref, min=0)
                errs = []
                for dt in (_sctc.BF16X3, None):
                    c = torch.full((M, N + 4), 7.0, device="cuda")
                    _gemm(L, _sctc, torch, dt, a, b, c, M, N, K, akc, bkc, bias, relu, use_ws)
                    assert float((c[:, N:] - 7.0).abs().max()) == 0.0, "wrote beyond N"
                    assert torch.isfinite(c[:, :N]).all()
                    errs.append(float(((c[:, :N].double() - ref).abs() / bound).max()))
                worst3, worst32 = max(worst3, errs[0]), max(worst32, errs[1])
                # 2^-23 per product from the dropped terms + fp32 accumulation: the same error class as
                # the fp32 fma chain, case by case (wide-range operands make sum|a||b| a loose scale)
                assert errs[0] < max(3.0 * errs[1], 2e-7), (M, N, K, akc, bkc, wide, errs)
    print("bf16x3 GEMM worst error / sum|a||b|: %.2e (fp32 matrix-core kernel on the same cases: %.2e)" % (worst3, worst32))
    assert worst3 < max(2.0 * worst32, 2e-7)


def test_gemm_bf16x3_odd_k_and_gather(mods):
    """weight-gradient layout (both operands [k][m]) with K not a multiple of the K tile, of 4 or of 2
    (a ragged minibatch's frame count), with and without split-K; K = 0 leaves bias / zero"""
    _sctc, _, _, torch = mods
    L = _sctc.lib()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for (M, N, K) in ((128, 128, 777), (260, 132, 1), (132, 388, 17), (1824, 512, 33), (516, 388, 4099), (64, 64, 0)):
        a = torch.randn((max(K, 1), M + 4), device="cuda")
        b = torch.randn((max(K, 1), N), device="cuda")
        for use_ws in (ws, None):
            c = torch.full((M, N + 4), 7.0, device="cuda")
            _gemm(L, _sctc, torch, _sctc.BF16X3, a, b, c, M, N, K, 0, 0, None, 0, use_ws)
            ref = a[:K, :M].double().t() @ b[:K, :N].double()
            bound = a[:K, :M].double().abs().t() @ b[:K, :N].double().abs() + 1e-30
            assert float((c[:, N:] - 7.0).abs().max()) == 0.0
            err = float(((c[:, :N].double() - ref).abs() / bound).max()) if K else float(c[:, :N].abs().max())
            # dropped split terms (<= 3 x 2^-24 per product) + fp32 accumulation of up to K = 4099 terms,
            # relative to sum|a||b|; observed over unseeded runs: up to 3.02e-7 (K = 33)
            assert err < 6e-7, (M, N, K, err)


def test_gemm_bf16x3_split_is_exact(mods):
    """operands that are exactly representable in ONE or TWO bfloat16 terms give the exact fp32
    product sums: integers up to 2^16 against +-1 -- every partial sum is an integer below 2^24"""
    _sctc, _, _, torch = mods
    L = _sctc.lib()
    M, N, K = 128, 128, 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    a = torch.randint(-65535, 65536, (M, K), device="cuda", generator=gen).float()
    b = (torch.randint(0, 2, (N, K), device="cuda", generator=gen) * 2 - 1).float()
    c = torch.empty((M, N), device="cuda")
    _gemm(L, _sctc, torch, _sctc.BF16X3, a, b, c, M, N, K, 1, 1, None, 0, None)
    assert torch.equal(c.double(), a.double() @ b.double().t())


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
def test_bf16x3_scaled_configs(mods, golden, name):
    """the reference-generated fixtures (rnnetcpu.py outputs) at the fp32 path's own tolerances"""
    _, brnnet, _, _ = mods
    params, grads, dims, data, labels, cost = load_net(golden("brnn_cfg.npz"), name + "_")
    net = make_net(brnnet, dims, params, gemm="bf16x3")
    c, g, skip = net.costAndGrad(data, labels)
    assert not skip
    assert c == pytest.approx(cost, rel=1e-4)
    w3 = check_grads(net, grads, dims[3])
    net32 = make_net(brnnet, dims, params)
    c32, _, _ = net32.costAndGrad(data, labels)
    w32 = check_grads(net32, grads, dims[3])
    print("%s: cost rel err bf16x3 %.1e / fp32 %.1e; worst gradient rel-norm error bf16x3 %.1e / fp32 %.1e"
          % (name, abs(c - cost) / cost, abs(c32 - cost) / cost, w3, w32))


def test_bf16x3_minibatch_masks_reg_and_accumulate(mods):
    """clip at 20 / strict mask / L2 term / skipped utterance / accumulate flag in a ragged
    minibatch: identical control flow, fp32-path tolerances against the oracle"""
    _, brnnet, obrnn, _ = mods
    rs = np.random.RandomState(15)
    D, A, H, NL, TL = 24, 9, 72, 4, 2
    Ts = [37, 12, 30, 5, 26, 37, 19]
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    params["Wf"] *= 1.5
    params["b"][TL - 1] += 4.0
    datas = [3.0 * rs.randn(D, T) for T in Ts]
    labs = [rs.randint(1, A, size=max(1, T // 6)).astype(np.int32) for T in Ts]
    labs[3] = np.array([4, 4, 4, 4], dtype=np.int32)          # infeasible at T=5 -> skipped
    with np.errstate(all="ignore"):
        c_ref, g_ref, s_ref, _ = obrnn.cost_and_grad_batch(params, datas, labs, TL, 20.0, 0.01)
    net = make_net(brnnet, (D, A, H, NL, TL, max(Ts)), params, maxUtts=len(Ts), reg=0.01, gemm="bf16x3")
    costs, _, skips = net.costAndGradBatch(datas, labs)
    np.testing.assert_array_equal(skips, s_ref)
    np.testing.assert_allclose(costs[~s_ref], c_ref[~s_ref], rtol=1e-4)
    check_grads(net, g_ref, NL)
    g1 = net.grad.flat.clone()
    net.costAndGradBatch(datas, labs, accumulate=True)
    import torch
    assert torch.allclose(net.grad.flat, 2 * g1, rtol=1e-5, atol=1e-6)
