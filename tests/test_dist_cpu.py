"""The N>1 path on CPU: world_size-2 `gloo` processes run the same gradient exchange the GPU
ranks run over RCCL (dist_sgd.allreduce_flat / DataParallel conventions), with per-utterance
gradients coming from the CPU oracle.  Checks SURVEY 8(e): all-reduced mean gradient on G ranks
== mean of the per-utterance oracle gradients; skipped utterances contribute zeros and are not
counted; every rank ends with identical buffers."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _flatten(g):
    parts = [w.ravel() for w in g["W"]] + [b.ravel() for b in g["b"]]
    if g["Wf"] is not None:
        parts += [g["Wf"].ravel(), g["Wb"].ravel()]
    return np.concatenate(parts)


def _problem():
    from oracle import brnn as obrnn
    rs = np.random.RandomState(42)
    D, A, H, NL, TL = 6, 5, 8, 2, 1
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    Ts = [9, 14, 5, 11, 7, 12, 10, 6]
    datas = [rs.randn(D, T) for T in Ts]
    labs = [rs.randint(1, A, size=2).astype(np.int32) for _ in Ts]
    labs[2] = np.array([3, 3, 3, 3], dtype=np.int32)        # T=5 < 7: infeasible -> skip
    return params, datas, labs, Ts, TL


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dist_sgd
    from oracle import brnn as obrnn
    with np.errstate(all="ignore"):
        params, datas, labs, Ts, TL = _problem()
        mine = dist_sgd.shard_utterances(Ts, world, rank)
        costs, g, skips, n_valid = obrnn.cost_and_grad_batch(
            params, [datas[i] for i in mine], [labs[i] for i in mine], TL)
    flat = torch.from_numpy(_flatten(g).astype(np.float32))
    side = torch.tensor([float(n_valid), float(costs[~skips].sum())], dtype=torch.float64)
    dist_sgd.allreduce_flat(flat, side, bucket_elems=37)    # odd bucket size: several buckets + tail
    out[rank] = (flat.numpy().copy(), side.numpy().copy(), mine)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_utterances_balances_frames():
    import dist_sgd
    Ts = [100, 90, 80, 70, 60, 50, 40, 30, 20, 10]
    shards = [dist_sgd.shard_utterances(Ts, 4, r) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(10))
    loads = [sum(Ts[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= 30
    assert dist_sgd.shard_utterances(Ts, 1, 0) == list(range(10))


@pytest.mark.timeout(300)
def test_gloo_world2_mean_gradient_matches_oracle():
    from oracle import brnn as obrnn
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    with np.errstate(all="ignore"):
        params, datas, labs, Ts, TL = _problem()
        costs, g, skips, n_valid = obrnn.cost_and_grad_batch(params, datas, labs, TL)
    assert skips.sum() == 1 and n_valid == len(Ts) - 1
    ref = _flatten(g)
    f0, s0, m0 = out[0]
    f1, s1, m1 = out[1]
    assert sorted(m0 + m1) == list(range(len(Ts)))
    np.testing.assert_array_equal(f0, f1)                   # identical on every rank
    np.testing.assert_array_equal(s0, s1)
    assert int(s0[0]) == n_valid                            # the skipped utterance is not counted
    assert s0[1] == pytest.approx(costs[~skips].sum(), rel=1e-12)
    np.testing.assert_allclose(f0, ref, rtol=2e-5, atol=1e-6)
    # mean-gradient convention of ctc/nnet.py:106-124
    np.testing.assert_allclose(f0 / s0[0], ref / n_valid, rtol=2e-5, atol=1e-6)


def _single_rank_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dist_sgd
    flat = torch.arange(1000, dtype=torch.float32) * 0.5
    side = torch.tensor([3.0, 7.5], dtype=torch.float64)
    ref = flat.clone()
    # default: a single rank has nothing to exchange and issues no collective
    os.environ.pop("SCTC_DIST_SINGLE_RANK", None)
    assert dist_sgd._nothing_to_reduce()
    dist_sgd.allreduce_flat(flat, side, bucket_elems=300)
    assert torch.equal(flat, ref)
    # SCTC_DIST_SINGLE_RANK=1: the collectives run (bucketed, asynchronous handles) and sum over one rank
    os.environ["SCTC_DIST_SINGLE_RANK"] = "1"
    assert not dist_sgd._nothing_to_reduce()
    dist_sgd.allreduce_flat(flat, side, bucket_elems=300)
    out.put((bool(torch.equal(flat, ref)), side.tolist()))
    dist.destroy_process_group()


def test_single_rank_collectives_switch():
    """dist_sgd on one rank: no collective by default, every collective of an N-rank run with
    SCTC_DIST_SINGLE_RANK=1 (the switch the 1-GPU RCCL tests use, tests/test_gpu_run.py)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(0, 1, _free_port(), out))
    p.start()
    same, side = out.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same and side == [3.0, 7.5]


# ---------------------------------------------------------------------------------------------
# the first real N > 1 launch, rehearsed without hardware

def _cfg4_lengths():
    """cfg-4's minibatch: 256 utterances, ragged, T_b ~ U[T/2, T] with T = 2000"""
    rs = np.random.RandomState(4)
    return [int(t) for t in rs.randint(1000, 2001, size=256)]


def test_shard_cfg4_minibatch_over_8_ranks_balances_frames():
    import dist_sgd
    Ts = _cfg4_lengths()
    for world in (2, 4, 8):
        shards = [dist_sgd.shard_utterances(Ts, world, r) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(len(Ts)))          # a partition
        loads = np.array([sum(Ts[i] for i in s) for s in shards], dtype=np.float64)
        assert (loads.max() - loads.min()) / loads.mean() <= 0.03, (world, loads)
        counts = [len(s) for s in shards]
        assert max(counts) - min(counts) <= 2, (world, counts)          # ~32 utterances per GPU at world 8
        # deterministic: every rank computes the same partition without talking to anybody
        assert shards == [dist_sgd.shard_utterances(list(Ts), world, r) for r in range(world)]


def _world8_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # what a launcher that gives every rank ONE visible device looks like from inside a rank
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ["HIP_VISIBLE_DEVICES"] = str(rank)
    os.environ.pop("SCTC_SHARED_DEVICE", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _sctc
    import dist_sgd
    from oracle import brnn as obrnn
    L = _sctc.lib()
    guess = bool(L.sctc_shared_device())      # what loading the library under this environment decided: nothing
    # (1) eight ranks, eight different GPUs (mocked bus ids): the device lease must stay OFF
    shared_a, ids_a = _sctc.resolve_shared_device(my_id=("node0", "0000:%02x:00.0" % (0x10 + rank)), log=False)
    mode_a = L.sctc_shared_device()
    # (2) ranks 2 and 5 on one physical GPU, everybody else alone: on for exactly those two
    bus = 0x10 + (2 if rank == 5 else rank)
    shared_b, ids_b = _sctc.resolve_shared_device(my_id=("node0", "0000:%02x:00.0" % bus), log=False)
    mode_b = L.sctc_shared_device()
    # (2b) the exchange only ever RAISES the mode: ranks 2 and 5, switched on above, stay on when a
    # later exchange sees eight distinct devices -- a timeout or a foreign process may have been the reason
    shared_b2, _ = _sctc.resolve_shared_device(my_id=("node0", "0000:%02x:00.0" % (0x10 + rank)), log=False)
    assert shared_b2 == (rank in (2, 5)) and bool(L.sctc_shared_device()) == (rank in (2, 5))
    L.sctc_set_shared_device(0)
    # (3) the same bus id on two different HOSTS is not sharing
    shared_c, _ = _sctc.resolve_shared_device(my_id=("node%d" % rank, "0000:10:00.0"), log=False)
    L.sctc_set_shared_device(0)
    # (4) the gradient exchange of a cfg-4-shaped minibatch (256 ragged utterances, scaled to a tiny net)
    with np.errstate(all="ignore"):
        params, datas, labs, Ts, TL = _problem256()
        mine = dist_sgd.shard_utterances(Ts, world, rank)
        costs, g, skips, n_valid = obrnn.cost_and_grad_batch(
            params, [datas[i] for i in mine], [labs[i] for i in mine], TL)
    flat = torch.from_numpy(_flatten(g).astype(np.float32))
    side = torch.tensor([float(n_valid), float(costs[~skips].sum())], dtype=torch.float64)
    dist_sgd.allreduce_flat(flat, side, bucket_elems=101)
    out[rank] = dict(guess=guess, a=(shared_a, mode_a, len(set(ids_a))), b=(shared_b, mode_b), c=shared_c,
                     flat=flat.numpy().copy(), side=side.numpy().copy(), mine=mine,
                     frames=sum(Ts[i] for i in mine), n_skip=int(skips.sum()))
    dist.barrier()
    dist.destroy_process_group()


def _problem256():
    from oracle import brnn as obrnn
    rs = np.random.RandomState(256)
    D, A, H, NL, TL = 5, 6, 8, 2, 1
    params = obrnn.init_params(D, A, H, NL, TL, rng=rs)
    Ts = [max(4, t // 100) for t in _cfg4_lengths()]            # 10 .. 20 frames
    datas = [rs.randn(D, T) for T in Ts]
    labs = [rs.randint(1, A, size=2).astype(np.int32) for _ in Ts]
    for i in (7, 100, 201):                                     # infeasible alignments -> skip
        labs[i] = np.array([3] * (Ts[i] // 2 + 2), dtype=np.int32)
    return params, datas, labs, Ts, TL


@pytest.mark.timeout(600)
def test_gloo_world8_bus_ids_and_mean_gradient():
    """world-size-8 rehearsal on CPU: (a) shared-device mode follows the PHYSICAL device identity the
    ranks exchange, not LOCAL_WORLD_SIZE vs the visible-device count (every rank sees one device here,
    the case that used to switch the lease on for all eight ranks of a real node); (b) the summed
    gradient / count / cost of cfg-4's 256 ragged utterances over 8 ranks equals one process's."""
    from oracle import brnn as obrnn
    world = 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world8_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [out[r] for r in range(world)]
    for r, o in enumerate(res):
        assert o["guess"] is False                              # HIP_VISIBLE_DEVICES set: no guess from counts
        assert o["a"] == (False, 0, world), (r, o["a"])         # 8 distinct GPUs: lease off on every rank
        assert o["b"] == ((r in (2, 5)), int(r in (2, 5))), (r, o["b"])
        assert o["c"] is False
    with np.errstate(all="ignore"):
        params, datas, labs, Ts, TL = _problem256()
        costs, g, skips, n_valid = obrnn.cost_and_grad_batch(params, datas, labs, TL)
    assert skips.sum() == 3 and n_valid == 253
    assert sum(o["n_skip"] for o in res) == 3
    assert sorted(sum((o["mine"] for o in res), [])) == list(range(256))
    frames = np.array([o["frames"] for o in res], dtype=np.float64)
    assert (frames.max() - frames.min()) / frames.mean() <= 0.03
    ref = _flatten(g)
    for o in res[1:]:
        np.testing.assert_array_equal(o["flat"], res[0]["flat"])     # identical on every rank
        np.testing.assert_array_equal(o["side"], res[0]["side"])
    assert int(res[0]["side"][0]) == n_valid
    assert res[0]["side"][1] == pytest.approx(costs[~skips].sum(), rel=1e-12)
    np.testing.assert_allclose(res[0]["flat"] / n_valid, ref / n_valid, rtol=3e-5, atol=1e-6)


def test_no_guess_from_environment_counts(monkeypatch):
    """loading the library never derives shared-device mode from LOCAL_WORLD_SIZE / visible-device
    counts (round 3 did, and an 8-GPU node whose launcher shows each rank one device looked shared);
    sharing is decided by physical identity: bus ids exchanged over the process group
    (shared_device_from_ids) or the library's per-device marker files at launch time"""
    import _sctc
    import torch as _t
    monkeypatch.setattr(_t.cuda, "device_count", lambda: 1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.delenv("SCTC_SHARED_DEVICE", raising=False)
    L = _sctc.lib()
    L.sctc_set_shared_device(0)
    assert L.sctc_shared_device() == 0
    assert _sctc.shared_device_from_ids([("a", "x"), ("a", "y"), ("a", "x")], 0)
    assert not _sctc.shared_device_from_ids([("a", "x"), ("a", "y"), ("a", "x")], 1)
    assert not _sctc.shared_device_from_ids([("a", "x"), ("b", "x")], 0)
