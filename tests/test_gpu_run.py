"""End-to-end driver test on the GPU: runNNet.run (train, checkpoint side files, resume) and
--test (writeLikelihoods: Kaldi BFM ark + pickle), on tiny synthetic shards in the reference's
on-disk format.  Checks the run-directory contract of ctc_fast/runNNet.py:143-205 and that the
cost goes down."""
import json
import os
import pickle

import numpy as np
import pytest

from tests.test_dataloader import write_shard

pytestmark = pytest.mark.gpu


def test_train_resume_and_export(tmp_path):
    import torch
    assert torch.cuda.is_available()
    import runNNet
    import writeLikelihoods as wl
    rs = np.random.RandomState(0)
    raw = img = 12
    data = tmp_path / "data"
    data.mkdir()
    A = 6
    for n in (1, 2):
        utts = [("f%d_u%d" % (n, i), int(rs.randint(12, 30)), list(rs.randint(1, A, size=3)))
                for i in range(6)]
        write_shard(data, n, utts, raw, rs)
    out = tmp_path / "run"
    common = ["--layerSize", "32", "--numLayers", "3", "--temporalLayer", "2", "--inputDim", str(img),
              "--rawDim", str(raw), "--outputDim", str(A), "--maxUttLen", "40", "--numFiles", "2",
              "--dataDir", str(data) + "/", "--step", "1e-3", "--momentum", "0.9", "--save_every", "1"]
    opt, nn = runNNet.run(common + ["--epochs", "1", "--outputDir", str(out)])
    assert (out / "epoch").read_text() == "0" and opt.it == 12
    assert opt.alpha == pytest.approx(1e-3 / 1.3)
    (out / "sentinel").unlink()
    # resume for a second epoch: state comes from params.pk, the step from step/anneal**epoch
    cfg = json.loads((out / "cfg.json").read_text())
    cfg["epochs"] = 2
    (out / "cfg.json").write_text(json.dumps(cfg))
    opt, nn = runNNet.run(["--cfg_file", str(out / "cfg.json")])
    assert opt.it == 24 and opt.alpha == pytest.approx(1e-3 / 1.3 ** 2)
    for name in ("cfg.json", "params.pk", "params.pk.epoch00", "params.pk.epoch01", "epoch", "num_files",
                 "last_cost", "sentinel", "train.log"):
        assert (out / name).exists(), name
    assert (out / "epoch").read_text() == "1"
    cfg = json.loads((out / "cfg.json").read_text())
    assert cfg["layerSize"] == 32 and cfg["param_count"] == nn._param_count
    with open(out / "params.pk", "rb") as f:
        it, costt, expcost, vel = pickle.load(f)
        stack = pickle.load(f)
    assert it == 24 and len(costt) == 24 and len(stack) == 3 + 3
    assert np.mean(costt[-6:]) < np.mean(costt[:6])               # it learns
    # test mode: log-likelihood export
    lik = tmp_path / "lik"
    runNNet.run(["--cfg_file", str(out / "cfg.json"), "--test", "--dataDir", str(data) + "/",
                 "--numFiles", "2", "--likDir", str(lik)])
    ark = wl.read_ark(str(lik / "loglikelihoods1.ark"))
    with open(lik / "loglikelihoods_1.pk", "rb") as f:
        pk = pickle.load(f)
    assert sorted(ark) == sorted(pk) and len(ark) == 6
    for k, m in ark.items():
        assert m.shape[1] == A and m.dtype == np.float32
        np.testing.assert_array_equal(m, pk[k].T)
        np.testing.assert_allclose(np.exp(m.astype(np.float64)).sum(axis=1), 1.0, rtol=1e-4)
    # the exported probabilities are those of the trained network
    import dataLoader as dl
    from nnets import brnnet
    loader = dl.DataLoader(str(data) + "/", raw, img)
    dd, _, keys, _ = loader.loadDataFileDict(1)
    net = brnnet.NNet(img, A, 32, 3, 40, train=False, temporalLayer=2)
    with open(out / "params.pk", "rb") as f:
        pickle.load(f)
        net.fromFile(f)
    p = net.costAndGrad(dd[keys[0]])
    np.testing.assert_allclose(np.log(p).T, ark[keys[0]], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("reg,backend", [(0.0, "gloo"), (0.02, "gloo"), (0.02, "nccl")])
def test_data_parallel_trainer_two_ranks(tmp_path, reg, backend):
    """(reg > 0: the L2 term must enter the update exactly once whatever the number of ranks --
    it is added after the all-reduce and the 1/n_valid scaling, never per rank; backend nccl =
    RCCL, one GPU per rank, runs wherever two devices are visible so that the driver's scaling
    run is not the first time RCCL executes this path)
    runNNet under torch.distributed.run with 2 ranks (gloo, both on the one GPU of the box):
    every rank processes its share of each minibatch, gradients are all-reduced, rank 0 owns the
    run directory -- and the parameters after one epoch equal those of the single-process run
    with the same minibatch (mean over the same utterances; fp32 summation order aside)."""
    import subprocess
    import sys
    import torch
    assert torch.cuda.is_available()
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank (%d visible)" % torch.cuda.device_count())
    import runNNet
    rs = np.random.RandomState(1)
    raw = img = 12
    data = tmp_path / "data"
    data.mkdir()
    A = 6
    utts = [("u%d" % i, int(rs.randint(12, 30)), list(rs.randint(1, A, size=3))) for i in range(8)]
    write_shard(data, 1, utts, raw, rs)
    common = ["--layerSize", "32", "--numLayers", "3", "--temporalLayer", "2", "--inputDim", str(img),
              "--rawDim", str(raw), "--outputDim", str(A), "--maxUttLen", "40", "--numFiles", "1",
              "--dataDir", str(data) + "/", "--step", "1e-3", "--momentum", "0.9", "--save_every", "1",
              "--epochs", "1", "--minibatch", "4", "--reg", str(reg)]
    single = tmp_path / "single"
    runNNet.run(common + ["--outputDir", str(single)])
    dp = tmp_path / "dp"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCTC_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.pathsep.join(
        [root, os.path.join(root, "stanford-ctc_amd"), os.environ.get("PYTHONPATH", "")]))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(29517 + (1 if reg else 0) + (2 if backend == "nccl" else 0)),
           os.path.join(root, "stanford-ctc_amd", "runNNet.py")] + common + ["--outputDir", str(dp)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert (dp / "sentinel").exists() and (dp / "train.log.rank1").exists()

    def load(path):
        with open(path, "rb") as f:
            state = pickle.load(f)
            return state, pickle.load(f)
    (it_s, cost_s, _, _), stack_s = load(single / "params.pk")
    (it_d, cost_d, _, _), stack_d = load(dp / "params.pk")
    assert it_s == it_d == 2
    np.testing.assert_allclose(cost_d, cost_s, rtol=1e-5)
    for (ws, bs), (wd, bd) in zip(stack_s, stack_d):
        np.testing.assert_allclose(wd, ws, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_two_ranks(backend):
    """bench.py under torch.distributed.run with 2 ranks (the driver's SCALE launch line): the
    overlapped per-layer all-reduce path end to end, on a reduced minibatch.  gloo: both ranks
    share the one GPU of the box; nccl (RCCL): only where two devices are visible."""
    import json
    import subprocess
    import sys
    import torch
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank (%d visible)" % torch.cuda.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCTC_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541" if backend == "gloo" else "29542",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "6",
           "--no-side", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["frames_per_step"] == 2 * 6 * 1000
    assert out["cost_check"]["rel_err"] < 1e-4
    # the ranks exchanged their PCI bus ids: the rank -> device map is in the line; two gloo ranks on the
    # one GPU of the box report the same id and take the device lease, RCCL ranks must sit on two devices
    ranks = out["config"]["rccl_ranks"]
    assert len(ranks) == 2 and out["config"]["backend"] == backend
    same = ranks[0].split(":", 1)[1] == ranks[1].split(":", 1)[1]
    assert out["config"]["shared_device_mode"] == same
    assert same == (backend == "gloo")


def test_bench_gpus2_without_a_launcher():
    """`python bench.py --gpus 2 ...` as the driver's N = 1 line is written -- no torch.distributed.run in the
    command, no RANK / WORLD_SIZE in the environment: the script starts the launcher itself (VERDICT r04 missing #4:
    the first SCALE run must not die on the WORLD_SIZE assert).  gloo: both ranks on the one GPU of the box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SCTC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "6", "--steps", "2", "--warmup", "1",
           "--no-side", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["frames_per_step"] == 2 * 6 * 1000 and out["cost_check"]["rel_err"] < 1e-4
    assert "without a launcher" in res.stderr
    # the N > 1 line explains itself (round 6): what the gradient exchange moved, how long it occupied the side stream,
    # how much of it was NOT hidden behind the backward pass, and the same per-GPU work on one GPU alone
    comm = out["comm"]
    assert comm["bytes"] > 4 * 20e6 and comm["buckets"] >= 7 and comm["allreduce_ms"] > 0
    assert 0 <= comm["exposed_ms"] <= comm["allreduce_ms"] + 1e-3 and comm["busbw_GBps"] > 0
    eff = out["scaling_efficiency_vs_n1"]
    assert eff["n1_ms"] > 0 and eff["n_ms"] > 0 and 0 < eff["value"] < 1.5
    # a failing rank's exit status comes back through the self-launch
    bad = subprocess.run(cmd[:2] + ["--gpus", "2", "--batch", "0", "--steps", "1", "--warmup", "0", "--no-side",
                                    "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def test_bench_four_ranks_gloo_on_one_gpu():
    """the driver's N = 4 launch line rehearsed on the one GPU of the box (gloo, all four ranks under the
    device lease): rank -> bus-id exchange over four ranks, four shards, MAX-over-ranks timing, one line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCTC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
           "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--batch", "3",
           "--no-side", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["value"] > 0 and out["config"]["frames_per_step"] == 4 * 3 * 1000
    assert out["cost_check"]["rel_err"] < 1e-4
    ranks = out["config"]["rccl_ranks"]
    assert len(ranks) == 4 and len({r.split(":", 1)[1] for r in ranks}) == 1      # one physical GPU
    assert out["config"]["shared_device_mode"] is True
    assert res.stderr.count("shared=1") == 4                                        # one line per rank


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_overlapped_allreduce_equals_flat_two_ranks(backend):
    """the per-layer bucketed all-reduce queued behind the engine's gradient events
    (dist_sgd.allreduce_overlapped) gives, bit for bit, what one flat all-reduce of the same local
    gradients gives -- 2 ranks with different utterances, 3 steps (tests/gpu_dist_equiv.py).
    gloo: both ranks on the one GPU of the box (the library's shared-device mode makes their
    persistent launches take turns); nccl (RCCL): wherever two devices are visible."""
    import json
    import subprocess
    import sys
    import torch
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank (%d visible)" % torch.cuda.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCTC_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29551" if backend == "gloo" else "29552",
           os.path.join(root, "tests", "gpu_dist_equiv.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["world"] == 2 and out["buckets"] == 6
    if backend == "gloo":
        assert out["shared_mode"] == 1        # two processes on one physical GPU: found through the marker files


def _rccl_one_rank_env(port):
    return dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1",
                MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                SCTC_DIST_SINGLE_RANK="1")


def _require_rccl_one_rank(port):
    """skip (not fail) when the box cannot create a one-rank RCCL communicator at all -- that is the
    environment's defect (tools/diag/rccl_one_rank_probe.py: init, one all-reduce, barrier);
    everything after a successful probe is asserted strictly"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _rccl_one_rank_env(port)
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "diag", "rccl_one_rank_probe.py")],
                         env=env, capture_output=True, text=True, timeout=300)
    if res.returncode != 0 or "barrier ok" not in res.stdout:
        pytest.skip("no one-rank RCCL communicator on this box: " + (res.stderr or res.stdout)[-300:])


def test_rccl_single_rank_overlapped_allreduce_equals_flat():
    """RCCL on the one GPU of the box: a ONE-rank `nccl` process group, with
    SCTC_DIST_SINGLE_RANK=1 so that dist_sgd issues every collective an N-rank run issues.  A
    one-rank all-reduce moves nothing over xGMI, but the RCCL communicator is created next to the
    engine, and ProcessGroupNCCL's own streams / work handles (which gloo's CPU-staged collectives
    do not have) order the per-layer buckets behind the engine's gradient events while the
    persistent BPTT grid runs: the result must equal the flat all-reduce and the local gradient
    bit for bit (tests/gpu_dist_equiv.py).  The two-rank twins above need two devices."""
    import json
    import subprocess
    import sys
    _require_rccl_one_rank(29560)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_rccl_one_rank_env(29561), SCTC_DIST_BACKEND="nccl")
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_dist_equiv.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["world"] == 1 and out["backend"] == "nccl" and out["buckets"] == 6
    assert out["shared_mode"] == 0            # one rank per device: no lease


def test_bench_single_rank_rccl():
    """bench.py's data-parallel step (asynchronous costAndGrad, per-layer RCCL all-reduces on the
    side stream, barrier + MAX over ranks) on a one-rank `nccl` group at the headline layer size:
    costs equal the single-GPU path's, and the collectives' bookkeeping does not slow the step down."""
    import json
    import subprocess
    import sys
    _require_rccl_one_rank(29564)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
            "--batch", "8", "--no-side", "--no-cpu-baseline"]
    outs = {}
    for mode in ("single", "dp"):
        env = _rccl_one_rank_env(29562)
        if mode == "dp":
            env.update(SCTC_BENCH_FORCE_DP="1", SCTC_BENCH_BACKEND="nccl")
        else:
            env.pop("SCTC_DIST_SINGLE_RANK")
        res = subprocess.run(base, env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
        outs[mode] = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    s, d = outs["single"], outs["dp"]
    assert d["config"]["parallelism"].startswith("dp1") and s["config"]["parallelism"] == "single-gpu"
    assert d["cost_mean"] == s["cost_mean"]                 # same kernels, same order: bit-identical costs
    assert d["cost_check"]["rel_err"] < 1e-4
    comm = d["comm"]                                        # one rank: real RCCL calls, nothing to move over a link
    assert comm["bytes"] > 4 * 20e6 and comm["buckets"] >= 7 and comm["allreduce_ms"] > 0 and comm["busbw_GBps"] == 0
    assert 0 <= comm["exposed_ms"] <= comm["allreduce_ms"] + 1e-3
    assert 0.7 < d["scaling_efficiency_vs_n1"]["value"] < 1.3 and "comm" not in s
    # bookkeeping only (a one-rank all-reduce moves no data): measured +1 %; the bound is loose because
    # three timed steps on a box shared with the harness are noisy
    assert d["ms_per_step"] < 1.15 * s["ms_per_step"] + 1.0, (d["ms_per_step"], s["ms_per_step"])


def test_c_level_allreduce_entry_one_rank_rccl():
    """sctc_brnn_allreduce_grads (round 5, SURVEY 8(b)'s sketched entry): the per-layer RCCL all-reduces queued behind
    the engine's gradient events by the LIBRARY, on a communicator the host created itself (ctypes on librccl.so,
    one rank: the sum over one rank must leave gradients and the side message bit for bit what the synchronous step
    gives), on a side stream while the backward pass is still queued; argument errors; a second step reuses it."""
    import ctypes
    import torch
    import _sctc
    from nnets import brnnet
    from oracle import brnn as obrnn
    rccl = ctypes.CDLL("librccl.so.1")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    L = _sctc.lib()
    rs = np.random.RandomState(12)
    D, A, H, NL, TL, B = 24, 20, 64, 3, 2, 7
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    Ts = [int(t) for t in rs.randint(3, 25, size=B)]
    datas = [rs.randn(D, T) for T in Ts]
    labs = [rs.randint(1, A, size=max(1, T // 6)).astype(np.int32) for T in Ts]
    from tests.test_gpu_brnn import host_stack
    net = brnnet.NNet(D, A, H, NL, max(Ts), temporalLayer=TL, maxUtts=B)
    net.setParams(host_stack(params))
    costs, _, skips = net.costAndGradBatch(datas, labs)
    want = net.grad.flat.clone()
    side = torch.tensor([3.0, 1.5, 0.25, 1.0], dtype=torch.float64, device="cuda")
    side_stream = torch.cuda.Stream()
    for step in range(2):
        net.grad.flat.zero_()
        cost_dev, skip_dev = net.costAndGradBatchAsync(datas, labs)
        rc = L.sctc_brnn_allreduce_grads(net._h, comm, _sctc.current_stream_ptr(), ctypes.c_void_p(side_stream.cuda_stream),
                                         ctypes.c_void_p(side.data_ptr()), 4, 1)
        _sctc.check(rc, "allreduce_grads")
        net.checkAsync()
        torch.cuda.synchronize()
        assert torch.equal(net.grad.flat, want)
        np.testing.assert_array_equal(cost_dev.cpu().numpy(), costs)
        np.testing.assert_array_equal(side.cpu().numpy(), [3.0, 1.5, 0.25, 1.0])
    # no backward pass queued (an empty shard): the zeroed buffer is reduced behind the compute stream as a whole
    net.grad.flat.zero_()
    _sctc.check(L.sctc_brnn_allreduce_grads(net._h, comm, _sctc.current_stream_ptr(), ctypes.c_void_p(side_stream.cuda_stream),
                                            None, 0, 0), "allreduce_grads")
    torch.cuda.synchronize()
    assert not net.grad.flat.any()
    with pytest.raises(ValueError):      # the compute stream is not a side stream
        _sctc.check(L.sctc_brnn_allreduce_grads(net._h, comm, _sctc.current_stream_ptr(), _sctc.current_stream_ptr(), None, 0, 1), "x")
    with pytest.raises(ValueError):
        _sctc.check(L.sctc_brnn_allreduce_grads(net._h, None, None, ctypes.c_void_p(side_stream.cuda_stream), None, 0, 1), "x")
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    rccl.ncclCommDestroy(comm)
