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
