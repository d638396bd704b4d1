"""NumPy float64 restatement of the reference BRNN step (TEST INFRASTRUCTURE).

Follows the reference's own CPU twin ``ctc_fast/debug-utils/rnnetcpu.py:54-150``
and adds what only the GPU model has (``ctc_fast/nnets/brnnet.py``):

* activation ceiling ``maxAct = 20.0`` on the recurrent layer
  (``brnnet.py:32,146-152``; rnnetcpu has no ceiling: pass ``max_act=None``),
* the strict ``0 < h < maxAct`` mask of ``within`` (``brnnet.py:208-209``;
  rnnetcpu uses ``np.sign`` at ``:125-126`` which is the same for h >= 0 without
  ceiling),
* L2 regularisation ``reg`` (``brnnet.py:178-183,197-198,244-247``).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
The CTC part is delegated to :mod:`oracle.ctc` (the C restatement), mirroring
``rnnetcpu.py:104``.

Layout: every matrix is (features, T) like the reference; weights are (out, in).
``params`` is the reference's ``stack`` flattened to
``{"W": [W1..W_{NL+1}], "b": [b1..b_{NL+1}], "Wf": .., "Wb": ..}``.
"""
import numpy as np

from . import ctc as octc


def init_params(input_dim, output_dim, layer_size, num_layers, temporal_layer, rng=np.random):
    """Reference initialisation (brnnet.py:38-41,66-70 == rnnetcpu.py:27-30,43-46):
    uniform +-sqrt(6)/sqrt(fan_in+fan_out); recurrent +-sqrt(6)/sqrt(2H); biases 0.
    Consumes ``rng.rand`` in the reference's order W1..W_{NL+1}, Wf, Wb so that
    ``np.random.seed(s)`` reproduces the reference's weights."""
    dims = [input_dim] + [layer_size] * num_layers + [output_dim]
    W, b = [], []
    for n_in, n_out in zip(dims[:-1], dims[1:]):
        s = np.sqrt(6.0) / np.sqrt(n_in + n_out)
        W.append(rng.rand(n_out, n_in) * 2 * s - s)
        b.append(np.zeros((n_out, 1)))
    p = {"W": W, "b": b, "Wf": None, "Wb": None}
    if 0 < temporal_layer < num_layers:          # brnnet.py:27-30
        s = np.sqrt(6.0) / np.sqrt(2.0 * layer_size)
        p["Wf"] = 2 * s * rng.rand(layer_size, layer_size) - s
        p["Wb"] = 2 * s * rng.rand(layer_size, layer_size) - s
    return p


def round_f16(x):
    """round-to-nearest-even to float16 and back (v_cvt_pk_f16_f32 of the mixed-precision path)"""
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float64)


def round_bf16(x):
    """round-to-nearest-even to bfloat16 and back (v_cvt_pk_bf16_f32)"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(np.shape(x))


class Mixed:
    """Numerics of the "fp16 activations" configuration (BASELINE configs[4]) restated: every
    time-batched contraction rounds BOTH operands to 16 bit (float16 in the forward pass,
    bfloat16 in the backward pass) and accumulates exactly; `rec` says whether the recurrent
    time step does the same (the 6..16-utterance kernel) or stays fp32 (the other kernels).
    Everything else -- biases, clips, masks, softmax, CTC -- is unrounded."""

    def __init__(self, rec=True):
        self.rec = rec
        self._cache = {}     # rounded weight matrices (the first operand of fwd / rec_*), by buffer

    def _w(self, w, fn):
        key = (fn.__name__, w.__array_interface__["data"][0], w.shape, w.strides)
        hit = self._cache.get(key)
        if hit is None:
            hit = (w, fn(w))                 # keeps `w` alive: the address stays unique
            self._cache[key] = hit
        return hit[1]

    def fwd(self, w, x):
        return self._w(w, round_f16) @ round_f16(x)

    def bwd(self, a, b):
        return round_bf16(a) @ round_bf16(b)

    def rec_fwd(self, w, x):
        return self._w(w, round_f16) @ round_f16(x) if self.rec else w @ x

    def rec_bwd(self, w, x):
        return self._w(w, round_bf16) @ round_bf16(x) if self.rec else w @ x


class _Exact:
    rec = False

    @staticmethod
    def fwd(a, b):
        return a @ b
    bwd = rec_fwd = rec_bwd = fwd


def _clip(x, max_act):
    x = np.maximum(x, 0.0)
    if max_act is not None:
        x = np.minimum(x, max_act)
    return x


def _open_mask(h, max_act):
    m = h > 0.0
    if max_act is not None:
        m &= h < max_act
    return m.astype(h.dtype)


def forward(params, data, temporal_layer, max_act=20.0, mixed=None):
    """Returns (logits, cache).  data: (D_in, T).  mixed: a Mixed() instance restates the
    16-bit-operand numerics, None = the reference's arithmetic."""
    mp = mixed or _Exact
    W, b = params["W"], params["b"]
    NL = len(W) - 1
    TL = temporal_layer if 0 < temporal_layer < NL else -1
    T = data.shape[1]
    acts = [np.asarray(data, dtype=np.float64)]
    hF = hB = preF = preB = None
    pre = [None]                      # pre-activations (before ReLU / clip): how close to a kink?
    for i in range(1, NL + 2):
        z = mp.fwd(W[i - 1], acts[i - 1]) + b[i - 1]
        pre.append(z)
        if i == TL:
            Wf, Wb = params["Wf"], params["Wb"]
            hF = np.zeros_like(z)
            hB = np.zeros_like(z)
            preF = np.array(z)
            preB = np.array(z)
            hF[:, 0] = _clip(z[:, 0], max_act)
            hB[:, T - 1] = _clip(z[:, T - 1], max_act)
            for t in range(1, T):
                preF[:, t] = z[:, t] + mp.rec_fwd(Wf, hF[:, t - 1])
                hF[:, t] = _clip(preF[:, t], max_act)
                u = T - 1 - t
                preB[:, u] = z[:, u] + mp.rec_fwd(Wb, hB[:, u + 1])
                hB[:, u] = _clip(preB[:, u], max_act)
            acts.append(hF + hB)
        elif i <= NL:
            acts.append(np.maximum(z, 0.0))
        else:
            acts.append(z)
    return acts[-1], {"acts": acts, "hF": hF, "hB": hB, "TL": TL, "NL": NL, "pre": pre,
                      "preF": preF, "preB": preB}


def softmax_cols(logits):
    e = np.exp(logits - logits.max(axis=0, keepdims=True))
    return e / e.sum(axis=0, keepdims=True)


def cost_and_grad(params, data, labels, temporal_layer, max_act=20.0, reg=0.0, blank=0, mixed=None,
                  masks=None, cache_out=None):
    """One utterance.  Returns (cost, grads, skip, probs) with ``grads`` shaped
    like ``params`` (dW list, db list, dWf, dWb).  On skip, grads is None (the
    reference returns its stale buffers, brnnet.py:185-186).
    Test instrumentation (not reference behaviour): ``masks`` = {"relu": {layer: (H,T) 0/1},
    "F": (H,T), "B": (H,T)} replaces the backward pass's own ReLU / (0,maxAct) masks -- a
    float32 device and this float64 restatement put a unit whose pre-activation is a rounding
    error away from a kink on different sides, and the two gradients then differ by that unit's
    whole delta; with the device's masks imposed, what is left is smooth arithmetic error.
    ``cache_out`` (a dict) receives the forward cache and "d1", the delta entering layer 1."""
    mp = mixed or _Exact
    W = params["W"]
    logits, cache = forward(params, data, temporal_layer, max_act, mixed)
    acts, hF, hB, TL, NL = cache["acts"], cache["hF"], cache["hB"], cache["TL"], cache["NL"]
    if cache_out is not None:
        cache_out.update(cache)
    T = data.shape[1]
    probs = softmax_cols(logits)
    cost, delta, skip = octc.ctc_loss(np.asfortranarray(probs),
                                      np.ascontiguousarray(labels, dtype=np.int32), blank)
    regcost = 0.0
    if reg > 0:                                           # brnnet.py:178-183
        mats = list(W) + ([params["Wf"], params["Wb"]] if TL > 0 else [])
        regcost = sum((reg / 2.0) * float(np.sum(m * m)) for m in mats)
        cost = cost + regcost
    if skip:
        return cost, None, True, probs

    dW = [None] * (NL + 1)
    db = [None] * (NL + 1)
    dWf = dWb = None
    d_in = np.array(delta)                                # (A, T)
    for i in range(NL, -1, -1):                           # brnnet.py:191-243
        dW[i] = mp.bwd(d_in, acts[i].T)
        if reg > 0:
            dW[i] = dW[i] + reg * W[i]
        db[i] = d_in.sum(axis=1, keepdims=True)
        if i == 0:
            break
        d_out = mp.bwd(W[i].T, d_in)
        if i == TL:
            Wf, Wb = params["Wf"], params["Wb"]
            mF, mB = _open_mask(hF, max_act), _open_mask(hB, max_act)
            if masks is not None:
                mF = np.asarray(masks["F"], dtype=np.float64)
                mB = np.asarray(masks["B"], dtype=np.float64)
            dF = np.array(d_out)
            dB = np.array(d_out)
            dF[:, T - 1] *= mF[:, T - 1]
            dB[:, 0] *= mB[:, 0]
            for t in range(1, T):
                u = T - 1 - t
                dF[:, u] = (dF[:, u] + mp.rec_bwd(Wf.T, dF[:, u + 1])) * mF[:, u]
                dB[:, t] = (dB[:, t] + mp.rec_bwd(Wb.T, dB[:, t - 1])) * mB[:, t]
            dWf = mp.bwd(dF[:, 1:], hF[:, :-1].T)         # brnnet.py:227-228
            dWb = mp.bwd(dB[:, :-1], hB[:, 1:].T)         # brnnet.py:229-230
            if reg > 0:
                dWf = dWf + reg * Wf
                dWb = dWb + reg * Wb
            d_out = dF + dB
        else:
            m = (acts[i] > 0.0) if masks is None else np.asarray(masks["relu"][i], dtype=np.float64)
            d_out = d_out * m                             # sign(h) for h >= 0, brnnet.py:236
        d_in = d_out
    if cache_out is not None:
        cache_out["d1"] = d_in
    grads = {"W": dW, "b": db, "Wf": dWf, "Wb": dWb}
    return cost, grads, False, probs


def cost_and_grad_batch(params, data_list, label_list, temporal_layer, max_act=20.0, reg=0.0,
                        mean=False, mixed=None):
    """Sum (or mean over non-skipped, ctc/nnet.py:106-124 convention) of
    per-utterance gradients -- the data-parallel parity target of SURVEY 8(e)."""
    total = None
    costs, skips = [], []
    for x, lab in zip(data_list, label_list):
        c, g, s, _ = cost_and_grad(params, x, lab, temporal_layer, max_act, reg=0.0, mixed=mixed)
        costs.append(c)
        skips.append(s)
        if s:
            continue
        if total is None:
            total = g
        else:
            total = {"W": [a + b for a, b in zip(total["W"], g["W"])],
                     "b": [a + b for a, b in zip(total["b"], g["b"])],
                     "Wf": None if g["Wf"] is None else total["Wf"] + g["Wf"],
                     "Wb": None if g["Wb"] is None else total["Wb"] + g["Wb"]}
    n_valid = sum(1 for s in skips if not s)
    if total is not None and mean and n_valid > 0:
        total = {"W": [a / n_valid for a in total["W"]], "b": [a / n_valid for a in total["b"]],
                 "Wf": None if total["Wf"] is None else total["Wf"] / n_valid,
                 "Wb": None if total["Wb"] is None else total["Wb"] / n_valid}
    if total is not None and reg > 0:
        total["W"] = [g + reg * w for g, w in zip(total["W"], params["W"])]
        if total["Wf"] is not None:
            total["Wf"] = total["Wf"] + reg * params["Wf"]
            total["Wb"] = total["Wb"] + reg * params["Wb"]
    return np.array(costs), total, np.array(skips), n_valid
