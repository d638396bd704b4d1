"""One rank of the overlapped-vs-flat all-reduce equivalence test (tests/test_gpu_run.py), run
under torch.distributed.run: a small BRNN step is queued without a host sync, the gradient buffer
is reduced (i) by dist_sgd.allreduce_overlapped -- per-layer buckets on a side stream behind the
engine's gradient events, while the backward pass still runs -- and (ii) by dist_sgd.allreduce_flat
on a copy of the same local gradients after a full sync.  Rank 0 prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stanford-ctc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    backend = os.environ.get("SCTC_DIST_BACKEND", "gloo")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    kw = {"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    import dist_sgd
    from nnets import brnnet
    D, A, H, NL, TL, T, B = 24, 12, 512, 3, 2, 30, 8
    np.random.seed(3)                                  # identical weights on every rank
    net = brnnet.NNet(D, A, H, NL, T, temporalLayer=TL, maxUtts=B)
    net.initParams()
    rs = np.random.RandomState(100 + rank)             # different utterances per rank
    Ts = [int(t) for t in rs.randint(5, T + 1, size=B)]
    datas = [rs.randn(D, t).astype(np.float32) for t in Ts]
    labs = [rs.randint(1, A, size=max(1, t // 8)).astype(np.int32) for t in Ts]
    worst = 0.0
    for step in range(3):
        # (ii) reference: full sync, then one flat all-reduce of a copy
        net.costAndGradBatch(datas, labs)
        local = net.grad.flat.clone()
        ref = local.clone()
        side_ref = torch.tensor([1.0, 2.0], dtype=torch.float64, device=ref.device)
        dist_sgd.allreduce_flat(ref, side_ref, bucket_elems=100003)
        # (i) the overlapped path on a freshly queued step
        cost_dev, skip_dev = net.costAndGradBatchAsync(datas, labs)
        side = torch.tensor([1.0, 2.0], dtype=torch.float64, device=ref.device)
        dist_sgd.allreduce_overlapped(net, side)
        net.checkAsync()
        torch.cuda.synchronize()
        got = net.grad.flat
        assert torch.equal(got, ref), "overlapped all-reduce differs from the flat one (step %d)" % step
        assert torch.equal(side, side_ref)
        if world > 1:
            assert not torch.equal(got, local)         # something was added
        else:
            assert torch.equal(got, local)             # one rank: the sum over ranks is the local gradient
        worst = max(worst, float((got - world * local).abs().max()))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"ok": True, "world": world, "backend": backend, "steps": 3,
                          "buckets": len(net.gradBuckets()), "shared_mode": int(__import__("_sctc").lib().sctc_shared_device())}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
