"""Data-parallel minibatch training across the GPUs of one node.

The reference has no multi-GPU path at all (one process = one GPU picked by CUDA_DEVICE,
runNNet.py:117-120; "distributed" = ssh-launched independent jobs, cluster/utils.py:112-128).
The only reference behaviour to match is the minibatch-mean convention of ctc/nnet.py:106-124:
the update uses the mean gradient over the non-skipped utterances.

Design (one process per GPU, torch.distributed; backend "nccl" IS RCCL on ROCm):
  * utterances shard by length-balanced round-robin (independent units, no data-path
    collective);
  * each rank runs the batched costAndGrad over its shard -> SUM of its utterances'
    gradients in one flat fp32 buffer (the engine's padded parameter layout);
  * ONE exchange step: all-reduce(sum) of the flat gradient buffer in a few large buckets
    (84 MB at cfg-3; xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so few large
    messages beat many small ones) plus a 2-float side message [n_valid, cost_sum];
  * every rank applies the identical update with grad_scale = 1/n_valid_global, so weights
    stay bit-identical without any broadcast after initialisation.
"""
import numpy as np

DEFAULT_BUCKET_ELEMS = 16 * 1024 * 1024     # 64 MB of fp32 per all-reduce call


def shard_utterances(lengths, world, rank):
    """indices of the utterances rank `rank` processes: longest first, each utterance dealt to
    the rank with the fewest frames so far (ties: lowest rank) -- balances sum(T) per rank
    (SURVEY 8(e) "Partitioning"); deterministic, so every rank computes the same partition."""
    lengths = np.asarray(lengths)
    order = np.argsort(-lengths, kind="stable")
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(int(i))
        load[r] += int(lengths[i])
    return shards[rank]


def _nothing_to_reduce(group=None):
    """a single-rank job has nothing to exchange -- unless SCTC_DIST_SINGLE_RANK=1 asks for the
    collectives anyway: a 1-GPU box can then drive the real backend (RCCL communicator,
    ProcessGroupNCCL's streams and work handles) through exactly the calls an N-rank run makes
    (tests/test_gpu_run.py::test_rccl_single_rank_*)."""
    import os
    import torch.distributed as dist
    if not dist.is_initialized():
        return True
    return dist.get_world_size(group) == 1 and os.environ.get("SCTC_DIST_SINGLE_RANK", "0") != "1"


def allreduce_flat(flat, side, bucket_elems=DEFAULT_BUCKET_ELEMS, group=None):
    """Sum-all-reduce the 1-D tensor `flat` in place, bucket by bucket, and the small 1-D
    float64 tensor `side` (e.g. [n_valid, cost_sum]).  Works for CUDA (RCCL) and CPU (gloo)
    tensors; asynchronous bucket handles are waited for at the end so that the copies of
    consecutive buckets overlap on the wire."""
    import torch.distributed as dist
    if _nothing_to_reduce(group):
        return flat, side
    handles = []
    n = flat.numel()
    for start in range(0, n, bucket_elems):
        handles.append(dist.all_reduce(flat[start:min(n, start + bucket_elems)],
                                       op=dist.ReduceOp.SUM, group=group, async_op=True))
    handles.append(dist.all_reduce(side, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for h in handles:
        h.wait()
    return flat, side


def allreduce_overlapped(net, side, group=None, side_stream=None, backward_queued=True, timing=None):
    """Sum-all-reduce net's flat gradient buffer bucket by bucket WHILE the backward pass queued
    on the current stream is still running: a side stream waits for the event the
    engine records when a layer's gradient is final (output layer first) and starts that layer's
    all-reduce; RCCL moves the finished layers over xGMI while the GEMMs / BPTT of the earlier
    layers compute.  `side` (small 1-D float64 device tensor, e.g. [n_valid, cost_sum, ...]) is
    reduced last.  On return the CURRENT stream has been made to wait for all of it.
    `backward_queued=False`: this rank queued no backward pass this step (empty shard; its gradient
    buffer was zeroed on the current stream instead) -- the engine's events are then stale or were
    never recorded, so the side stream is ordered behind the current stream as a whole.
    `timing`: a dict that receives three timing events of this exchange -- "first_start" (side stream, the first
    bucket's gradient is final: its all-reduce starts), "last_end" (side stream, every collective has finished),
    "backward_end" (current stream, behind the backward pass as queued so far) -- plus "bytes" and "buckets";
    comm_stats(timing) turns them into milliseconds once the streams have been synchronised."""
    import torch
    import torch.distributed as dist
    import _sctc
    if _nothing_to_reduce(group):
        return
    L = _sctc.lib()
    flat = net.grad.flat
    cur = torch.cuda.current_stream()
    st = side_stream or torch.cuda.Stream()
    works = []
    if timing is not None:
        timing.clear()
        timing["backward_end"] = torch.cuda.Event(enable_timing=True)
        timing["backward_end"].record(cur)
        timing["bytes"] = 0
        timing["buckets"] = 0
    with torch.cuda.stream(st):
        if not backward_queued:
            st.wait_stream(cur)                    # the zero fill, not a stale event, orders the buckets
        for ev, start, end in net.gradBuckets():
            if backward_queued:
                _sctc.check(L.sctc_stream_wait_event(st.cuda_stream, ev), "stream_wait_event")
            if timing is not None:
                if "first_start" not in timing:
                    timing["first_start"] = torch.cuda.Event(enable_timing=True)
                    timing["first_start"].record(st)
                timing["bytes"] += (end - start) * flat.element_size()
                timing["buckets"] += 1
            works.append(dist.all_reduce(flat[start:end], op=dist.ReduceOp.SUM, group=group,
                                         async_op=True))
        st.wait_stream(cur)                        # `side` is produced on the compute stream
        works.append(dist.all_reduce(side, op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()                               # the side stream waits for the collectives
        if timing is not None:
            timing["bytes"] += side.numel() * side.element_size()
            timing["buckets"] += 1
            timing["last_end"] = torch.cuda.Event(enable_timing=True)
            timing["last_end"].record(st)
    cur.wait_stream(st)


def comm_stats(timing, world):
    """milliseconds and bandwidths of one overlapped exchange (allreduce_overlapped(..., timing=...)), after its streams
    have been synchronised: allreduce_ms = first bucket's start -> last collective's end (the span the exchange occupies
    on the side stream, waits for later layers' gradients included); exposed_ms = how far that end lies BEHIND the end
    of the backward pass on the compute stream (the part of the exchange that is not hidden; 0 when it finishes first);
    algbw = bytes / allreduce_ms; busbw = algbw x 2 (N - 1) / N (the ring / tree-independent figure of nccl-tests)."""
    if not timing or "last_end" not in timing or "first_start" not in timing:
        return None
    span = timing["first_start"].elapsed_time(timing["last_end"])
    exposed = max(0.0, timing["backward_end"].elapsed_time(timing["last_end"]))
    algbw = timing["bytes"] / (span * 1e-3) / 1e9 if span > 0 else 0.0
    return {"bytes": int(timing["bytes"]), "buckets": int(timing["buckets"]), "allreduce_ms": span,
            "exposed_ms": exposed, "algbw_GBps": algbw, "busbw_GBps": algbw * 2.0 * (world - 1) / max(1, world)}


class DataParallel(object):
    """Wraps an nnets.brnnet.NNet whose gradient stack lives in one flat device buffer."""

    def __init__(self, net, bucket_elems=DEFAULT_BUCKET_ELEMS, group=None):
        self.net = net
        self.bucket_elems = bucket_elems
        self.group = group
        self._side_stream = None
        self.time_comm = False      # record timing events around the overlapped exchange (bench.py: the `comm` field)
        self._timing = {}
        self.n_valid = 0
        self.cost_sum = 0.0
        self.regcost = 0.0
        # which ranks sit on which physical GPU: the ranks exchange (hostname, PCI bus id) -- two on
        # one GPU must take the device lease around their persistent recurrent launches, eight on eight
        # GPUs must NOT (it would drain the stream around every launch and serialise the overlapped
        # all-reduce), whatever HIP_VISIBLE_DEVICES / LOCAL_WORLD_SIZE look like
        # (a COLLECTIVE over `group`: every rank constructs its DataParallel at the same point; the device is the one
        # the net's gradient buffer lives on, not whatever the current device happens to be)
        import _sctc
        dev = getattr(getattr(net, "grad", None), "flat", None)
        dev = dev.device.index if dev is not None and getattr(dev, "is_cuda", False) else None
        self.shared_device, self.device_ids = _sctc.resolve_shared_device(group, device=dev)

    def allreduce_gradients(self, n_valid_local, cost_sum_local=0.0, regcost_local=None):
        """after net.costAndGradBatch on every rank: sums gradients, utterance counts and costs
        over the ranks.  `regcost_local`: this rank's L2 cost if it evaluated the model this step
        (None for a rank whose shard was empty); self.regcost becomes the value of the ranks
        that did (identical weights, so identical values).  Returns grad_scale =
        1/n_valid_global (0 if every utterance skipped)."""
        import torch
        flat = self.net.grad.flat
        side = torch.tensor([float(n_valid_local), float(cost_sum_local),
                             0.0 if regcost_local is None else float(regcost_local),
                             0.0 if regcost_local is None else 1.0], dtype=torch.float64,
                            device=flat.device)
        allreduce_flat(flat, side, self.bucket_elems, self.group)
        return self._finish(side)

    def _finish(self, side):
        side = side.cpu()
        self.n_valid = int(round(side[0].item()))
        self.cost_sum = float(side[1].item())
        self.regcost = float(side[2].item() / side[3].item()) if side[3].item() > 0 else 0.0
        return 1.0 / self.n_valid if self.n_valid > 0 else 0.0

    def allreduce_gradients_overlapped(self, cost_dev, skip_dev, regcost_local=None):
        """after net.costAndGradBatchAsync (nothing synchronised yet): queues the per-layer
        all-reduces behind the engine's gradient events on a side stream, builds the side message
        [n_valid, cost_sum, regcost, has_regcost] ON THE DEVICE from the cost / skip arrays, and
        only then synchronises.  cost_dev / skip_dev None = this rank's shard was empty (its
        gradient buffer must already be zero)."""
        import torch
        flat = self.net.grad.flat
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        side = torch.zeros(4, dtype=torch.float64, device=flat.device)
        if cost_dev is not None:
            valid = skip_dev == 0
            side[0] = valid.sum()
            side[1] = torch.where(valid, cost_dev, torch.zeros_like(cost_dev)).sum()
        if regcost_local is not None:       # 0-dim device tensor (NNet.regCostDev) or a number
            side[2] = regcost_local
            side[3] = 1.0
        allreduce_overlapped(self.net, side, self.group, self._side_stream,
                             backward_queued=cost_dev is not None, timing=self._timing if self.time_comm else None)
        return self._finish(side)

    def comm_stats(self):
        """timings of the last overlapped exchange (see comm_stats above); needs time_comm = True before the step and
        a synchronised device (allreduce_gradients_overlapped ends in one); None if nothing was exchanged"""
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        return comm_stats(self._timing, world)
