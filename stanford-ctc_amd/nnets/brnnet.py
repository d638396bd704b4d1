"""Drop-in for ``ctc_fast/nnets/brnnet.py``: ``NNet`` -- a deep ReLU network in which
one hidden layer is bi-directionally recurrent, trained with CTC -- running on the
MI355X through libsctc_hip.so (include/sctc.h).  Same constructor, methods and
attributes as the reference class (brnnet.py:8-277); what the methods launch is new:

* every ``cm.dot`` becomes one fp32 MFMA GEMM over all frames of the minibatch,
* the 2(T-1) ``mvdot_col_slice`` + ``minmax`` launches of the temporal layer
  (brnnet.py:148-152) and their BPTT mirror (:215-224) become one persistent
  weight-stationary kernel per pass,
* softmax + CTC run on the device (no D2H/H2D of probs/deltas, brnnet.py:170,188).

``costAndGrad(data, labels)`` keeps the reference's one-utterance semantics
(returns ``(cost, grad, skip)``; ``grad`` is the model-owned gradient stack, stale on
skip like brnnet.py:185-186).  ``costAndGradBatch`` is the minibatch extension
(gradients summed over utterances).  There is no CPU fallback.
"""
import ctypes
import pickle

import os

import numpy as np

import _sctc
import cudamat as cm


class TensorStack(list):
    """list of [w, b] CUDAMatrix pairs that are views into one flat device buffer"""
    flat = None


class NNet:

    def __init__(self, inputDim, outputDim, layerSize, numLayers, maxBatch,
                 train=True, temporalLayer=-1, reg=0.0, maxUtts=1, fp16=False, gemm=None):
        cm.cublas_init()                      # brnnet.py:13 (fails here without the library)
        self.outputDim = outputDim
        self.inputDim = inputDim
        self.layerSize = layerSize
        self.numLayers = numLayers
        self.layerSizes = [layerSize] * numLayers
        self.maxBatch = maxBatch
        self.maxUtts = maxUtts
        # extension (BASELINE configs[4] "fp16 acts / fp32 alpha-beta"): 16-bit operands on the
        # matrix cores, fp32 accumulation; False = the reference's fp32 arithmetic
        self.fp16 = bool(fp16)
        # extension: which matrix-core instruction carries the fp32 time-batched contractions.
        # None / "f32": v_mfma_f32_32x32x2_f32 (the reference's fp32 fma arithmetic);
        # "bf16x3": every fp32 operand split exactly into three bfloat16 terms, six cross products
        # on the bfloat16 matrix cores, fp32 accumulation -- fp32-accurate, same tolerances
        if gemm is None:                      # drop-in switch for unchanged callers (runNNet.py, sgd.py)
            gemm = os.environ.get("SCTC_GEMM") or None
        if gemm not in (None, "f32", "bf16x3"):
            raise ValueError("gemm must be None, 'f32' or 'bf16x3'")
        if gemm == "bf16x3" and self.fp16:
            raise ValueError("gemm='bf16x3' is an fp32 mode; it excludes fp16=True")
        self.gemm = gemm or "f32"
        self.train = train
        self.reg = reg
        self.regcost = 0.0
        if not self.train:
            np.seterr(all='ignore')           # brnnet.py:25-26
        if temporalLayer <= 0 or temporalLayer >= numLayers:
            self.temporalLayer = -1           # brnnet.py:27-30
        else:
            self.temporalLayer = temporalLayer
        if reg > 0 and numLayers + 1 > 128:
            raise ValueError("reg > 0 supports at most 127 layers (bias ranges of the fused L2 step)")
        self.maxAct = 20.0                    # brnnet.py:32
        self._h = None
        self.stack = None
        self.grad = None

    # ------------------------------------------------------------------ set-up

    def _config(self):
        return _sctc.BrnnConfig(self.inputDim, self.outputDim, self.layerSize, self.numLayers,
                                self.temporalLayer, int(self.maxBatch) * int(self.maxUtts),
                                int(self.maxUtts), float(self.maxAct) if self.maxAct else 0.0,
                                float(self.reg), 1 if self.train else 0,
                                _sctc.F16 if self.fp16 else
                                (_sctc.BF16X3 if self.gemm == "bf16x3" else _sctc.F32))

    def _allocate(self):
        torch = _sctc.require_gpu()
        L = _sctc.lib()
        cfg = self._config()
        sizes = _sctc.BrnnSizes()
        _sctc.check(L.sctc_brnn_query(ctypes.byref(cfg), ctypes.byref(sizes)), "NNet")
        self._cfg = cfg
        self._params = torch.zeros(sizes.param_elems, dtype=torch.float32, device="cuda")
        self._grads = (torch.zeros(sizes.param_elems, dtype=torch.float32, device="cuda")
                       if self.train else None)
        self._ws = torch.empty(sizes.workspace_bytes, dtype=torch.uint8, device="cuda")
        h = ctypes.c_void_p()
        rc = L.sctc_brnn_create(ctypes.byref(cfg), self._params.data_ptr(),
                                self._grads.data_ptr() if self.train else None,
                                self._ws.data_ptr(), sizes.workspace_bytes, ctypes.byref(h))
        _sctc.check(rc, "NNet")
        self._h = h
        self._param_count = int(sizes.param_count)
        infos = []
        for i in range(sizes.n_tensors):
            ti = _sctc.TensorInfo()
            _sctc.check(L.sctc_brnn_tensor_info(h, i, ctypes.byref(ti)), "tensor_info")
            infos.append(ti)
        self._infos = infos
        self.stack = self._make_stack(self._params)
        if self.train:
            self.grad = self._make_stack(self._grads)

    def _make_stack(self, flat):
        """[W1,b1]..[W_{NL+1},b_{NL+1}] (+[Wf,dummy],[Wb,dummy]), brnnet.py:58-59,71-72"""
        torch = _sctc.require_gpu()
        st = TensorStack()
        st.flat = flat
        mats = []
        for ti in self._infos:
            rows_p, ld = cm.padded_layout(ti.rows, ti.cols)
            mats.append(cm.CUDAMatrix(_flat=flat[ti.offset:ti.offset + rows_p * ld],
                                      _shape=(ti.rows, ti.cols)))
        n_ff = 2 * (self.numLayers + 1)
        for i in range(0, n_ff, 2):
            st.append([mats[i], mats[i + 1]])
        if self.temporalLayer > 0:
            dummy = cm.empty((1, 1))          # shared zero "bias" of the temporal layer (:61-64)
            st.append([mats[n_ff], dummy])
            st.append([mats[n_ff + 1], dummy])
        return st

    def initParams(self):
        """Initialize parameters using 6/sqrt(fanin+fanout) (brnnet.py:34-41,66-70).
        Draws from np.random in the reference's order, so np.random.seed(s) gives the
        reference's weights."""
        sizes = [self.inputDim] + self.layerSizes + [self.outputDim]
        scales = [np.sqrt(6) / np.sqrt(n + m) for n, m in zip(sizes[:-1], sizes[1:])]
        host = [[np.random.rand(m, n) * 2 * s - s, np.zeros((m, 1))]
                for n, m, s in zip(sizes[:-1], sizes[1:], scales)]
        if self.temporalLayer > 0:
            scale = np.sqrt(6) / np.sqrt(self.layerSize * 2)
            host.append([2 * scale * np.random.rand(self.layerSize, self.layerSize) - scale, None])
            host.append([2 * scale * np.random.rand(self.layerSize, self.layerSize) - scale, None])
        self._allocate()
        self.setParams(host)

    def setParams(self, host_stack):
        """host_stack: list of [w, b] NumPy arrays in stack order (b may be None)"""
        if self._h is None:
            self._allocate()
        for (w, b), (hw, hb) in zip(self.stack, host_stack):
            w.numpy_array = np.asarray(hw, dtype=np.float32)
            w.copy_to_device()
            if hb is not None and b.shape == np.asarray(hb).reshape(-1, 1).shape:
                b.numpy_array = np.asarray(hb, dtype=np.float32).reshape(-1, 1)
                b.copy_to_device()

    def paramCount(self):
        param_count = 0
        for w, b in self.stack:
            print(w.shape, b.shape)
            param_count += np.prod(w.shape)
            param_count += np.prod(b.shape)
        return param_count

    def setViews(self, batchSize):
        """The reference slices its maxBatch-wide buffers here (brnnet.py:96-115); the
        engine's workspace needs no views, only the capacity check remains."""
        assert batchSize <= self.maxBatch, "Batch size exceeds max batch"

    # ------------------------------------------------------------------ the hot path

    def _stage(self, data_list):
        """host (inputDim, T) arrays -> one device float32 [sum T][inputDim] block"""
        torch = _sctc.require_gpu()
        rows = []
        for d in data_list:
            d = np.asarray(d)
            if d.ndim != 2 or d.shape[0] != self.inputDim:
                raise ValueError("data must be (inputDim, T); got %s" % (d.shape,))
            # the loader's (inputDim, T) Fortran-ordered float32 views are [T][inputDim] in memory:
            # no host copy; anything else is converted once
            rows.append(np.ascontiguousarray(d.T, dtype=np.float32))
        if len(rows) == 1:
            return torch.from_numpy(rows[0]).cuda(non_blocking=True)
        # one device block, one copy per utterance straight from where the features live
        # (a DMA when that is the loader's page-locked shard buffer) -- no 60 MB host concatenate
        dev = torch.empty((sum(r.shape[0] for r in rows), self.inputDim), dtype=torch.float32,
                          device="cuda")
        o = 0
        for r in rows:
            dev[o:o + r.shape[0]].copy_(torch.from_numpy(r), non_blocking=True)
            o += r.shape[0]
        return dev

    def _grow_ctc_workspace(self, mb, h=None):
        """The reference allocates its lattices per call, (2U+1) x T, with no bound (ctc_fast.pyx:22-32); the model's
        workspace reserves 2048 lattice states per frame.  A minibatch with a longer label row gets a CTC scratch of its
        own size here (kept, and only ever grown), instead of SCTC_ERR_WORKSPACE -- rounds 1-5 made the trainer skip
        such utterances."""
        torch = _sctc.require_gpu()
        need, have = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _sctc.check(_sctc.lib().sctc_brnn_ctc_workspace_bytes(h or self._h, ctypes.byref(mb), ctypes.byref(need),
                                                              ctypes.byref(have)), "ctc workspace")
        if need.value > have.value:
            torch.cuda.synchronize()             # a step that still uses the previous scratch may be in flight
            ws = torch.empty(need.value + need.value // 8, dtype=torch.uint8, device="cuda")
            _sctc.check(_sctc.lib().sctc_brnn_set_ctc_workspace(h or self._h, ws.data_ptr(), ws.numel()), "ctc workspace")
            if h is None or h == self._h:
                self._ctc_ws = ws
            else:
                self._lane_ctc_ws = getattr(self, "_lane_ctc_ws", {})
                self._lane_ctc_ws[int(getattr(h, "value", h) or 0)] = ws

    def _minibatch(self, feats_dev, T_b, labels_list):
        T_arr = np.ascontiguousarray(T_b, dtype=np.int32)
        keep = [T_arr]
        if labels_list is not None:
            U_arr = np.ascontiguousarray([len(l) for l in labels_list], dtype=np.int32)
            lab = np.ascontiguousarray(np.concatenate([np.asarray(l).reshape(-1)
                                                       for l in labels_list]), dtype=np.int32)
            keep += [U_arr, lab]
            mb = _sctc.Minibatch(len(T_arr), _sctc.i32(T_arr), feats_dev.data_ptr(),
                                 _sctc.i32(lab), _sctc.i32(U_arr))
            if self.train and U_arr.size and int(U_arr.max()) > 1023 and self._h is not None:
                self._grow_ctc_workspace(mb)     # a label row beyond the 2048 states the workspace reserves per frame
        else:
            mb = _sctc.Minibatch(len(T_arr), _sctc.i32(T_arr), feats_dev.data_ptr(), None, None)
        return mb, keep

    def costAndGradBatch(self, data_list, labels_list, sync_skip=False, accumulate=False,
                         feats_dev=None, T_b=None, reg_in_grad=True):
        """Minibatch step.  Returns (costs float64[B], grad stack, skips bool[B]); the
        gradient is the SUM over the non-skipped utterances.  `feats_dev`/`T_b` let the
        caller pass features that already sit in HBM ([sum T][inputDim] float32).
        With reg > 0 the L2 term reg*W is added once per CALL (brnnet.py:197-198 adds it
        once per utterance = per call); minibatch / data-parallel trainers pass
        reg_in_grad=False and apply it once after the all-reduce and the 1/n_valid scaling
        (sgd.py here).  The costs never contain the L2 cost; self.regcost holds it."""
        if self._h is None:
            raise RuntimeError("initParams() / fromFile() first")
        if feats_dev is None:
            T_b = [np.asarray(d).shape[1] for d in data_list]
            feats_dev = self._stage(data_list)
        for T in T_b:
            self.setViews(T)
        mb, keep = self._minibatch(feats_dev, T_b, labels_list)
        B = len(T_b)
        cost = np.zeros(B, dtype=np.float64)
        skip = np.zeros(B, dtype=np.int32)
        regcost = ctypes.c_double(0.0)
        flags = (_sctc.FLAG_SYNC_SKIP if sync_skip else 0) | \
                (_sctc.FLAG_ACCUMULATE if accumulate else 0) | \
                (0 if reg_in_grad else _sctc.FLAG_NO_REG_GRAD)
        rc = _sctc.lib().sctc_brnn_cost_and_grad(
            self._h, ctypes.byref(mb), flags, cost.ctypes.data_as(_sctc.c_f64p),
            skip.ctypes.data_as(_sctc.c_i32p), ctypes.byref(regcost), _sctc.current_stream_ptr())
        _sctc.check(rc, "costAndGrad")
        if self.reg > 0:
            self.regcost = regcost.value            # brnnet.py:178-183
        return cost, self.grad, skip.astype(bool)

    def costAndGradBatchAsync(self, data_list, labels_list, accumulate=False, feats_dev=None,
                              T_b=None, reg_in_grad=True):
        """costAndGradBatch without a host sync: everything is queued on the current stream and
        the per-utterance costs / skip flags stay on the device (float64 [B], int32 [B]).  Call
        checkAsync() before trusting the results.  Used by the data-parallel trainer so that the
        per-layer gradient all-reduces can be queued while the backward pass still runs."""
        torch = _sctc.require_gpu()
        if self._h is None:
            raise RuntimeError("initParams() / fromFile() first")
        if feats_dev is None:
            T_b = [np.asarray(d).shape[1] for d in data_list]
            feats_dev = self._stage(data_list)
        for T in T_b:
            self.setViews(T)
        mb, keep = self._minibatch(feats_dev, T_b, labels_list)
        B = len(T_b)
        cost_dev = torch.empty(B, dtype=torch.float64, device="cuda")
        skip_dev = torch.empty(B, dtype=torch.int32, device="cuda")
        flags = (_sctc.FLAG_ACCUMULATE if accumulate else 0) | \
                (0 if reg_in_grad else _sctc.FLAG_NO_REG_GRAD)
        rc = _sctc.lib().sctc_brnn_cost_and_grad_async(
            self._h, ctypes.byref(mb), flags, cost_dev.data_ptr(), skip_dev.data_ptr(),
            _sctc.current_stream_ptr())
        _sctc.check(rc, "costAndGrad")
        self._async_keep = (feats_dev, keep)      # host label arrays / features outlive the launches
        return cost_dev, skip_dev

    # ------------------------------------------------------------------ one utterance per stream

    def _lanes_for(self, n_streams):
        """2 x n_streams minibatch-1 engines over the SAME parameter buffer (two per HIP stream, used
        alternately: an engine's host-side plan must not be rewritten while its previous utterance
        is still queued), each with its own workspace and (lane 0 excepted: it writes the model's
        gradient stack directly) gradient buffer"""
        torch = _sctc.require_gpu()
        lanes = getattr(self, "_lanes", None)
        if lanes is not None and len(lanes) == 2 * n_streams:
            return lanes
        L = _sctc.lib()
        for ln in lanes or []:                         # a different stream count: drop the old engines
            L.sctc_brnn_destroy(ln["h"])
        self._lanes = None
        cfg = _sctc.BrnnConfig(self.inputDim, self.outputDim, self.layerSize, self.numLayers,
                               self.temporalLayer, int(self.maxBatch), 1,
                               float(self.maxAct) if self.maxAct else 0.0, float(self.reg), 1,
                               self._cfg.operand_dtype)
        sizes = _sctc.BrnnSizes()
        _sctc.check(L.sctc_brnn_query(ctypes.byref(cfg), ctypes.byref(sizes)), "NNet lanes")
        lanes = []
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        for k in range(2 * n_streams):
            grads = self._grads if k == 0 else torch.zeros_like(self._grads)
            ws = torch.empty(sizes.workspace_bytes, dtype=torch.uint8, device="cuda")
            h = ctypes.c_void_p()
            _sctc.check(L.sctc_brnn_create(ctypes.byref(cfg), self._params.data_ptr(), grads.data_ptr(),
                                           ws.data_ptr(), sizes.workspace_bytes, ctypes.byref(h)), "NNet lanes")
            lanes.append({"h": h, "grads": grads, "ws": ws, "stream": streams[k % n_streams], "cfg": cfg})
        self._lanes = lanes
        return lanes

    def costAndGradStreams(self, data_list, labels_list, n_streams=2, feats_dev=None, T_b=None,
                           reg_in_grad=True):
        """north_star's "a minibatch of utterances shards one-utterance-per-stream on one GPU": every
        utterance is its own minibatch-1 costAndGrad -- the reference's mode, sgd.py:70-95 -- queued
        on one of `n_streams` HIP streams without a host sync; the streams' kernels overlap on the
        device (two 228-workgroup grids of the small-batch persistent recurrence fit the part side by
        side; the library's in-process gate admits no more than fit, recurrent.hip).  Returns what
        costAndGradBatch returns: (costs float64[B], grad stack = SUM over the non-skipped
        utterances, skips bool[B]).  Measured slower than the packed minibatch (bench.py
        `one_utterance_per_stream`); kept because it is the variant the north star names."""
        torch = _sctc.require_gpu()
        if self._h is None:
            raise RuntimeError("initParams() / fromFile() first")
        if not self.train:
            raise RuntimeError("costAndGradStreams needs a train=True model")
        if feats_dev is None:
            T_b = [np.asarray(d).shape[1] for d in data_list]
            feats_dev = self._stage(data_list)
        for T in T_b:
            self.setViews(T)
        B = len(T_b)
        L = _sctc.lib()
        lanes = self._lanes_for(max(1, min(int(n_streams), B)))
        cur = torch.cuda.current_stream()
        cost_dev = torch.zeros(B, dtype=torch.float64, device="cuda")
        skip_dev = torch.zeros(B, dtype=torch.int32, device="cuda")
        keep = []
        used = [False] * len(lanes)
        for ln in lanes[:len(lanes) // 2]:
            ln["stream"].wait_stream(cur)              # features / parameters are produced on `cur`
        off = 0
        for i in range(B):
            k = i % len(lanes)
            ln = lanes[k]
            mb, kp = self._minibatch(feats_dev[off:off + T_b[i]], [T_b[i]], [labels_list[i]])
            keep.append((mb, kp))
            off += T_b[i]
            if len(labels_list[i]) > 1023 and ln["h"] is not self._h:
                self._grow_ctc_workspace(mb, h=ln["h"])   # this lane's engine has a workspace of its own
            flags = (_sctc.FLAG_ACCUMULATE if used[k] else 0) | \
                    (0 if (reg_in_grad and i == 0) else _sctc.FLAG_NO_REG_GRAD)
            if used[k]:
                # an engine's host-side plan (row tables, CTC descriptors) is reused by its next call:
                # the engine's previous utterance (two rounds ago on this stream) must have consumed it
                _sctc.check(L.sctc_brnn_check(ln["h"], ctypes.c_void_p(ln["stream"].cuda_stream)), "costAndGrad")
            used[k] = True
            rc = L.sctc_brnn_cost_and_grad_async(ln["h"], ctypes.byref(mb), flags,
                                                 cost_dev.data_ptr() + 8 * i, skip_dev.data_ptr() + 4 * i,
                                                 ctypes.c_void_p(ln["stream"].cuda_stream))
            _sctc.check(rc, "costAndGrad")
        for k, ln in enumerate(lanes):
            cur.wait_stream(ln["stream"])
            if not used[k]:                            # fewer utterances than lanes: this engine ran nothing
                continue
            _sctc.check(L.sctc_brnn_check(ln["h"], ctypes.c_void_p(ln["stream"].cuda_stream)), "costAndGrad")
            if k > 0:                                  # lane 0 wrote the model's gradient stack itself
                _sctc.check(L.sctc_axpy(self._grads.data_ptr(), ln["grads"].data_ptr(), 1.0,
                                        self._grads.numel(), _sctc.current_stream_ptr()), "costAndGrad")
        if self.reg > 0:
            self.regcost = float(self.regCostDev().item())
        cost = cost_dev.cpu().numpy()
        skip = skip_dev.cpu().numpy().astype(bool)
        del keep
        return cost, self.grad, skip

    def checkAsync(self):
        """synchronises the current stream; raises if a persistent kernel of the queued step timed out"""
        _sctc.check(_sctc.lib().sctc_brnn_check(self._h, _sctc.current_stream_ptr()), "costAndGrad")

    def recurrentPath(self):
        """(forward, bptt, retries): which recurrence the last step ran -- 1 one persistent launch
        per pass, 2 the same under the inter-process device lease, 3 one launch per time step (the
        non-persistent fallback) -- and how many steps of this model were re-run after a timeout"""
        f, b, r = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _sctc.check(_sctc.lib().sctc_brnn_recurrent_path(self._h, ctypes.byref(f), ctypes.byref(b),
                                                         ctypes.byref(r)), "recurrentPath")
        return f.value, b.value, r.value

    def debugBuffer(self, which):
        """diagnostics: host copy (float64, [rows][cols]) of an internal matrix of the last call --
        which: 0..numLayers = hActs[i], 100 / 101 = hActsFor / hActsBack, 102 = the temporal layer's W h + b, 200 = delta entering layer 1"""
        torch = _sctc.require_gpu()
        ptr, rows, cols, ld = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _sctc.check(_sctc.lib().sctc_brnn_debug_buffer(self._h, which, ctypes.byref(ptr), ctypes.byref(rows),
                                                       ctypes.byref(cols), ctypes.byref(ld)), "debugBuffer")
        torch.cuda.synchronize()
        off = (ptr.value - self._ws.data_ptr()) // 4
        flat = self._ws.view(torch.float32)[off:off + rows.value * ld.value]
        return flat.view(rows.value, ld.value)[:, :cols.value].cpu().numpy().astype(np.float64)

    def gradBuckets(self):
        """[(event, start, end)] over the flat gradient buffer in the order the backward pass
        finishes them (output layer first, brnnet.py:191-193; the recurrent pair right after the
        temporal layer's BPTT); `event` is the raw hipEvent_t recorded when [start, end) is final"""
        L = _sctc.lib()
        NL = self.numLayers
        offs = [int(ti.offset) for ti in self._infos] + [int(self._grads.numel())]
        out = []
        for i in range(NL, -1, -1):
            out.append((L.sctc_brnn_grad_event(self._h, 2 * i), offs[2 * i], offs[2 * i + 2]))
            if i == self.temporalLayer:
                for k in (2 * (NL + 1), 2 * (NL + 1) + 1):
                    out.append((L.sctc_brnn_grad_event(self._h, k), offs[k], offs[k + 1]))
        return out

    def regCostDev(self):
        """(reg/2) * sum ||w||^2 over the weight tensors (brnnet.py:178-183) for the CURRENT
        parameters as a 0-dim float64 device tensor (no host sync)"""
        torch = _sctc.require_gpu()
        L = _sctc.lib()
        if not hasattr(self, "_reg_ws"):
            self._reg_ws = torch.empty(8192, dtype=torch.uint8, device="cuda")
        mats = [ti for ti in self._infos if ti.kind != 1]
        out = torch.zeros(len(mats), dtype=torch.float64, device="cuda")
        for j, ti in enumerate(mats):
            rows_p, ld = cm.padded_layout(ti.rows, ti.cols)
            _sctc.check(L.sctc_sumsq(self._params.data_ptr() + 4 * int(ti.offset), rows_p * ld,
                                     out.data_ptr() + 8 * j, self._reg_ws.data_ptr(),
                                     self._reg_ws.numel(), _sctc.current_stream_ptr()), "regcost")
        return 0.5 * float(self.reg) * out.sum()

    def costAndGrad(self, data, labels=None, sentence=None):
        T = data.shape[1]
        self.setViews(T)
        if not self.train:
            return self.forwardProbs([data])[0]     # brnnet.py:171-173
        cost, grad, skip = self.costAndGradBatch([data], [labels], sync_skip=True)
        c = float(cost[0])
        if self.reg > 0:
            c = c + self.regcost                    # added even when skipped (brnnet.py:178-186)
        return c, self.grad, bool(skip[0])

    def forwardProbs(self, data_list):
        """train=False path for a list of utterances -> list of float32 (outputDim, T) probs"""
        torch = _sctc.require_gpu()
        if self._h is None:
            raise RuntimeError("initParams() / fromFile() first")
        T_b = [np.asarray(d).shape[1] for d in data_list]
        feats = self._stage(data_list)
        mb, keep = self._minibatch(feats, T_b, None)
        out = torch.empty((int(sum(T_b)), self.outputDim), dtype=torch.float32, device="cuda")
        rc = _sctc.lib().sctc_brnn_forward(self._h, ctypes.byref(mb), out.data_ptr(),
                                           _sctc.current_stream_ptr())
        _sctc.check(rc, "costAndGrad(train=False)")
        host = out.cpu().numpy()
        res, o = [], 0
        for T in T_b:
            res.append(np.asfortranarray(host[o:o + T].T))
            o += T
        return res

    def updateParams(self, scale, update):
        """w += scale*dw, b += scale*db for every pair (brnnet.py:251-256).  When `update`
        is a stack over one flat buffer this is a single launch."""
        if isinstance(update, TensorStack) and update.flat is not None \
                and update.flat.numel() == self._params.numel():
            _sctc.check(_sctc.lib().sctc_axpy(self._params.data_ptr(), update.flat.data_ptr(),
                                              float(scale), self._params.numel(),
                                              _sctc.current_stream_ptr()), "updateParams")
            return
        for params, paramsDel in zip(self.stack, update):
            w, b = params
            dw, db = paramsDel
            w.add_mult(dw, alpha=scale)
            b.add_mult(db, alpha=scale)

    def zerosLikeStack(self):
        """a fresh stack (e.g. the SGD velocity, sgd.py:21-23) over one flat buffer"""
        torch = _sctc.require_gpu()
        return self._make_stack(torch.zeros_like(self._params))

    # ------------------------------------------------------------------ checkpoint (brnnet.py:258-277)

    def toFile(self, fid):
        """Saves only the network parameters to the given fd."""
        stack = []
        for w, b in self.stack:
            w.copy_to_host()
            b.copy_to_host()
            stack.append([w.numpy_array, b.numpy_array])
        pickle.dump(stack, fid)

    def noreg_ranges(self):
        """[beg, end) element ranges of the flat parameter buffer that hold biases (no L2 term)"""
        r = []
        for ti in self._infos:
            if ti.kind == 1:
                r += [int(ti.offset), int(ti.offset) + cm.padded_layout(ti.rows, ti.cols)[0]]
        return np.ascontiguousarray(r, dtype=np.int64)

    def fromFile(self, fid):
        # encoding='latin1': params.pk written by the reference is a Python-2 cPickle of NumPy
        # arrays (brnnet.py:258-267), which Python 3 can only read this way
        stack = pickle.load(fid, encoding='latin1')
        if self._h is None:
            self._allocate()
        for (w, b), (wi, bi) in zip(self.stack, stack):
            w.numpy_array = np.array(wi, dtype=np.float32)
            b.numpy_array = np.array(bi, dtype=np.float32).reshape(b.shape)
            w.copy_to_device()
            b.copy_to_device()

    def check_grad(self, data, labels, epsilon=1e-3):
        """forward-difference gradient check on a corner of every weight (brnnet.py:279-297)"""
        cost, grad, _ = self.costAndGrad(data, labels)
        worst = 0.0
        for param, delta in zip(self.stack, grad):
            w, b = param
            dw, db = delta
            dw.copy_to_host()
            w.copy_to_host()
            analytic = dw.numpy_array.copy()
            for i in range(min(w.shape[0], 3)):
                for j in range(min(w.shape[1], 3)):
                    w.numpy_array[i, j] += epsilon
                    w.copy_to_device()
                    costP, _, _ = self.costAndGrad(data, labels)
                    numGrad = (costP - cost) / epsilon
                    w.numpy_array[i, j] -= epsilon
                    w.copy_to_device()
                    print("Analytic %f, Numeric %f" % (analytic[i, j], numGrad))
                    worst = max(worst, abs(analytic[i, j] - numGrad))
        return worst

    def __del__(self):
        try:
            for ln in getattr(self, "_lanes", None) or []:
                _sctc.lib().sctc_brnn_destroy(ln["h"])
            self._lanes = None
            if self._h is not None:
                _sctc.lib().sctc_brnn_destroy(self._h)
                self._h = None
        except Exception:
            pass
