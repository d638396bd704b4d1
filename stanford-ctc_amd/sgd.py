"""Drop-in for ``ctc_fast/sgd.py``: class ``SGD`` -- Nesterov momentum with global
gradient-norm clipping -- driving ``nnets.brnnet.NNet`` on the MI355X.

Same constructor, ``run(data_dict, alis, keys, sizes)``, ``toFile``/``fromFile`` and
bookkeeping attributes (``it``, ``costt``, ``expcost``, ``regcost``) as the reference
(sgd.py:8-167).  What changes underneath: the reference spends ~13
passes over the 84 MB parameter set and 14 host syncs per utterance (three ``updateParams``
sweeps, 14 ``euclid_norm()`` round trips, 28 ``mult``/``add_mult`` launches); here the
look-ahead / undo are one axpy each over the flat parameter buffer, the norm is one
two-stage reduction and the velocity+weight update is ONE fused kernel that applies the
clipping factor on the device (``sctc_nesterov_step``).

Extensions (off by default, reference behaviour is ``minibatch=1`` on one GPU):
``minibatch=N`` processes N utterances per step through ``costAndGradBatch`` and, under
``torch.distributed``, shards them over the ranks and all-reduces the gradient
(mean over non-skipped utterances, ctc/nnet.py:106-124 convention; dist_sgd.py).
"""
import logging
import pickle
import random

import numpy as np

import _sctc
import dist_sgd

LATTICE_STATES_RESERVED = 2048  # lattice row the engine's workspace reserves per frame (csrc/brnn_engine.hip CTC_LP_MAX)


def lattice_fits(T, U, max_frames):
    """Whether an utterance's CTC scratch fits the share of the workspace the model was created with (2048 lattice states
    for each of its maxBatch frames).  Round 5 made the trainer skip utterances for which this is False; since round 6
    nnets.brnnet grows the CTC scratch for such a minibatch by itself (sctc_brnn_ctc_workspace_bytes /
    sctc_brnn_set_ctc_workspace), like the reference, which allocates its lattices per call with no bound
    (ctc_fast.pyx:22-32) -- the trainer no longer asks.  Kept as a statement of what needs no extra allocation."""
    L = 2 * U + 1
    if L <= LATTICE_STATES_RESERVED:
        return True
    w = (L + 1 + 63) // 64 * 64
    return (T + 2) * w <= (max_frames + 2) * LATTICE_STATES_RESERVED


class SGD:

    def __init__(self, model, maxBatch, alpha=1e-2, optimizer='nesterov',
                 momentum=0.9, maxGradNorm=1500, minibatch=1):
        self.model = model
        self.maxBatch = maxBatch
        self.it = 0
        self.momentum = momentum
        self.alpha = alpha
        self.optimizer = optimizer
        self.maxGNorm = maxGradNorm
        self.minibatch = int(minibatch)
        if self.optimizer == 'nesterov':
            # sgd.py:21-23 -- one flat buffer with the model's layout
            self.velocity = self.model.zerosLikeStack()
        elif self.optimizer == 'adagrad':
            assert False                    # dead branch in the reference as well (sgd.py:24-26)
        else:
            raise ValueError("unknown optimizer %r" % (optimizer,))
        self.costt = []
        self.expcost = []
        self.regcost = []
        torch = _sctc.require_gpu()
        self._sumsq = torch.zeros(1, dtype=torch.float64, device="cuda")
        self._ws = torch.empty(8192, dtype=torch.uint8, device="cuda")
        self._dp = None
        self.last_gnorm = 0.0

    # ------------------------------------------------------------------ checkpoint (sgd.py:36-55)

    def toFile(self, fid):
        stack = []
        for w, b in self.velocity:
            w.copy_to_host()
            b.copy_to_host()
            stack.append([w.numpy_array, b.numpy_array])
        pickle.dump([self.it, self.costt, self.expcost, stack], fid)

    def fromFile(self, fid):
        # latin1: the reference writes this with Python-2 cPickle (sgd.py:36-42)
        it, costt, expcost, stack = pickle.load(fid, encoding='latin1')
        self.it = it
        self.costt = costt
        self.expcost = expcost
        for (w, b), (wi, bi) in zip(self.velocity, stack):
            w.numpy_array = np.array(wi, dtype=np.float32)
            b.numpy_array = np.array(bi, dtype=np.float32).reshape(b.shape)
            w.copy_to_device()
            b.copy_to_device()

    # ------------------------------------------------------------------ device-side step

    def _grad_sumsq(self, grad_scale=1.0, reg=0.0, mom=0.0):
        """|| effective gradient ||^2 on the device.  reg == 0: sum g^2 (the caller scales the norm
        by grad_scale); reg > 0 (minibatch / data-parallel): sum (grad_scale*g + reg*(w + mom*v))^2
        with the L2 term entering exactly once, after the all-reduce, evaluated at the look-ahead
        point like the reference's (sgd.py:91-95, brnnet.py:197-198; biases excluded, :197-200)"""
        g = self.model.grad.flat
        if reg > 0.0:
            nr = self.model.noreg_ranges()
            _sctc.check(_sctc.lib().sctc_sumsq_reg(
                g.data_ptr(), self.model._params.data_ptr(), self.velocity.flat.data_ptr(), float(mom),
                float(grad_scale), float(reg),
                g.numel(), _sctc.i64(nr), len(nr) // 2, self._sumsq.data_ptr(),
                self._ws.data_ptr(), self._ws.numel(), _sctc.current_stream_ptr()), "euclid_norm")
            return
        _sctc.check(_sctc.lib().sctc_sumsq(g.data_ptr(), g.numel(), self._sumsq.data_ptr(),
                                           self._ws.data_ptr(), self._ws.numel(),
                                           _sctc.current_stream_ptr()), "euclid_norm")

    def _apply(self, mom, grad_scale, reg=0.0):
        """velocity = mom*velocity - alph*grad ; params += velocity  (sgd.py:129-141,161) with
        alph = alpha * min(1, maxGNorm/gnorm) evaluated on the device"""
        m = self.model
        if reg > 0.0:
            nr = m.noreg_ranges()
            _sctc.check(_sctc.lib().sctc_nesterov_step_reg(
                m._params.data_ptr(), self.velocity.flat.data_ptr(), m.grad.flat.data_ptr(),
                m._params.numel(), float(mom), float(self.alpha), float(self.maxGNorm),
                float(grad_scale), float(reg), _sctc.i64(nr), len(nr) // 2,
                self._sumsq.data_ptr(), _sctc.current_stream_ptr()), "SGD step")
            return
        _sctc.check(_sctc.lib().sctc_nesterov_step(
            m._params.data_ptr(), self.velocity.flat.data_ptr(), m.grad.flat.data_ptr(),
            m._params.numel(), float(mom), float(self.alpha), float(self.maxGNorm),
            float(grad_scale), self._sumsq.data_ptr(), _sctc.current_stream_ptr()), "SGD step")

    def _bookkeep(self, cost):
        if np.isfinite(cost):
            # compute exponentially weighted cost (sgd.py:113-126)
            if self.it > 1 and len(self.expcost) > 0:
                self.expcost.append(.01 * cost + .99 * self.expcost[-1])
            else:
                self.expcost.append(cost)
            self.costt.append(cost)
            if self.model.reg > 0.0:
                rc = self.model.regcost
                if len(self.regcost) > 0:
                    self.regcost.append(0.01 * rc + 0.99 * self.regcost[-1])
                else:
                    self.regcost.append(rc)

    # ------------------------------------------------------------------ the loop (sgd.py:57-167)

    def run(self, data_dict, alis, keys, sizes=None):
        """Runs stochastic gradient descent with nesterov acceleration.  Model is objective."""
        momIncrease = 10
        mom = 0.5
        # randomly select minibatch (shuffles the caller's list in place, like sgd.py:68)
        random.shuffle(keys)
        usable = []
        for k in keys:
            if self.minibatch <= 1:
                # the reference counts every key, also the ones it filters out (sgd.py:70-80)
                self.it += 1
            mb_data = data_dict[k]
            if mb_data.shape[1] > self.maxBatch:
                logging.info("SKIPPING utt exceeds batch length (Utterance length %d)."
                             % mb_data.shape[1])
                continue
            mb_labels = np.array(alis[k], dtype=np.int32)
            if mb_data.shape[1] < mb_labels.shape[0]:
                logging.info("SKIPPING utt frames less than label length (Utterance length %d, "
                             "Num Labels %d)." % (mb_data.shape[1], mb_labels.shape[0]))
                continue
            # conditions the engine rejects (the reference would index out of bounds): one bad utterance must not end
            # the run.  (A label row beyond the 2048 lattice states the workspace reserves per frame is NOT one of them
            # any more: the model grows its CTC scratch, round 6.)
            if mb_labels.shape[0] < 1 or mb_labels.min() < 0 or mb_labels.max() >= self.model.outputDim:
                logging.info("SKIPPING utt with unusable labels (Num Labels %d, ids %s..%s)."
                             % (mb_labels.shape[0], mb_labels.min() if mb_labels.size else '-',
                                mb_labels.max() if mb_labels.size else '-'))
                continue
            if self.minibatch <= 1:
                self._step([(k, mb_data, mb_labels)], momIncrease, mom)
            else:
                usable.append((k, mb_data, mb_labels))
        for i in range(0, len(usable), max(1, self.minibatch)):
            self.it += 1
            self._step(usable[i:i + self.minibatch], momIncrease, mom)

    def _step(self, batch, momIncrease, mom):
        import torch.distributed as dist
        if self.it > momIncrease:
            mom = self.momentum
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        if world > 1:
            mine = dist_sgd.shard_utterances([d.shape[1] for _, d, _ in batch], world, rank)
            local = [batch[j] for j in mine]
            if self._dp is None:
                self._dp = dist_sgd.DataParallel(self.model)
        else:
            local = batch
        m = self.model
        # minibatch / data-parallel: the engine returns the SUM of the data gradients; the L2
        # term enters once, after the all-reduce and the 1/n_valid scaling (_grad_sumsq/_apply)
        reference_mode = len(batch) == 1 and world == 1
        reg_late = 0.0 if reference_mode else float(m.reg)
        # w = w + mom*velocity (evaluate gradient at future point), sgd.py:91-93
        m.updateParams(mom, self.velocity)
        cost_dev = skip_dev = regc = None
        try:
            if reference_mode:
                k, data, labels = batch[0]
                cost, grad, skip = m.costAndGrad(data, labels)
                if m.reg > 0:
                    cost -= m.regcost        # bookkept below like the batched path
                costs, skips = np.array([cost]), np.array([skip])
            elif world > 1:
                # nothing synchronises here: the step is queued, then the per-layer all-reduces
                # are queued behind the engine's gradient events (dist_sgd.allreduce_overlapped)
                if local:
                    cost_dev, skip_dev = m.costAndGradBatchAsync([d for _, d, _ in local],
                                                                 [l for _, _, l in local],
                                                                 reg_in_grad=False)
                    if m.reg > 0:
                        regc = m.regCostDev()        # at the look-ahead point, like brnnet.py:178
                else:
                    m.grad.flat.zero_()
            elif local:
                costs, grad, skips = m.costAndGradBatch([d for _, d, _ in local],
                                                        [l for _, _, l in local],
                                                        reg_in_grad=False)
        finally:
            # undo update: w = w - mom*velocity, sgd.py:97-100 (also when the engine raised)
            m.updateParams(-mom, self.velocity)
        if world > 1:
            self._dp.allreduce_gradients_overlapped(cost_dev, skip_dev, regc)
            m.checkAsync()
            n_valid, cost_sum = self._dp.n_valid, self._dp.cost_sum
            if m.reg > 0:
                m.regcost = self._dp.regcost
        else:
            n_valid = int((~skips).sum())
            cost_sum = float(np.sum(costs[~skips])) if n_valid else 0.0
        if n_valid == 0:
            for (k, d, l) in batch:
                logging.info("SKIPPING: Key=%s, SeqLen=%d, NumFrames=%d." % (k, l.shape[0], d.shape[1]))
            return
        grad_scale = 1.0 / n_valid      # mean over non-skipped utterances (1.0 for the reference's B=1)
        # Compute norm of all parameters as one vector (sgd.py:102-107): one reduction
        self._grad_sumsq(grad_scale, reg_late, mom)
        # the reference's cost includes the L2 cost (brnnet.py:178-183); same convention for
        # every minibatch size (rank-local regcost is identical on all ranks)
        cost = cost_sum / n_valid + (m.regcost if m.reg > 0 else 0.0)
        self._bookkeep(cost)
        self._apply(mom, grad_scale, reg_late)
        self.last_gnorm = float(np.sqrt(self._sumsq.item())) * (1.0 if reg_late > 0 else grad_scale)
        if rank == 0:
            k, d, l = batch[0]
            logging.info("Iter %d : Cost=%.4f, ExpCost=%.4f, GradNorm=%.4f, SeqLen=%d, NumFrames=%d."
                         % (self.it, cost, self.expcost[-1] if self.expcost else float('nan'),
                            self.last_gnorm, l.shape[0], sum(x[1].shape[1] for x in batch)))
