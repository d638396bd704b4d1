"""Forward-only export: runs a ``train=False`` NNet over a feature shard
and writes the per-frame log-probabilities the reference's decoders consume --
``ctc_fast/analysis-utils/writeLikelihoods.py:8-55``:

* ``loglikelihoods<N>.ark``: per utterance ``<key> `` + a Kaldi binary float-matrix header
  (``\\0B`` ``FM `` ``\\4`` int32 rows ``\\4`` int32 cols) + float32 ``[frames][outputDim]`` rows,
* ``loglikelihoods_<N>.pk``: pickle of ``{key: float32 (outputDim, frames) log-probs}``.

All utterances of a shard go through the network as ONE minibatch-free sequence of batched
forward calls (``NNet.forwardProbs``), not one launch chain per utterance.
"""
import os
import pickle
import struct

import numpy as np


def kaldi_matrix_header(key, rows, cols):
    """bytes of ``key`` + the binary FM header (writeLikelihoods.py:8-26)"""
    return (key.encode() + b" " + struct.pack("b", 0) + b"BFM " + struct.pack("b", 4) +
            struct.pack("i", int(rows)) + struct.pack("b", 4) + struct.pack("i", int(cols)))


def read_ark(path):
    """inverse of the writer (used by the tests): {key: float32 (cols, rows)... as written}"""
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        sp = data.index(b" ", pos)
        key = data[pos:sp].decode()
        pos = sp + 1
        assert data[pos:pos + 6] == b"\x00BFM \x04", "not a binary float matrix"
        rows = struct.unpack("i", data[pos + 6:pos + 10])[0]
        assert data[pos + 10:pos + 11] == b"\x04"
        cols = struct.unpack("i", data[pos + 11:pos + 15])[0]
        pos += 15
        n = rows * cols * 4
        out[key] = np.frombuffer(data[pos:pos + n], dtype=np.float32).reshape(rows, cols)
        pos += n
    return out


def writeLogLikes(loader, nn, fn, outDir, writePickle=False, batch=32):
    data_dict, alis, keys, sizes = loader.loadDataFileDict(fn)
    lik_dict = {}
    with np.errstate(divide="ignore"):
        with open(os.path.join(outDir, "loglikelihoods%d.ark" % fn), "wb") as fid:
            group = max(1, min(int(batch), int(getattr(nn, "maxUtts", 1))))
            for g in range(0, len(keys), group):
                ks = keys[g:g + group]
                for k in ks:
                    assert data_dict[k].shape[1] < nn.maxBatch, "Need larger max utt length."
                probs_list = nn.forwardProbs([data_dict[k] for k in ks])
                for k, probs in zip(ks, probs_list):
                    assert probs.dtype == np.float32, "Probs array malformed."
                    assert probs.shape[0] == nn.outputDim, "Probs dimensions mismatch."
                    logp = np.log(probs)
                    fid.write(kaldi_matrix_header(k, probs.shape[1], nn.outputDim))
                    np.ascontiguousarray(logp.T).tofile(fid)
                    lik_dict[k] = logp
    if writePickle:
        with open(os.path.join(outDir, "loglikelihoods_%d.pk" % fn), "wb") as f:
            pickle.dump(lik_dict, f)
    return lik_dict
