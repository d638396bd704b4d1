"""Drop-in for ``ctc_fast/dataLoader.py``: reader of the Kaldi-exported
training shards the reference trains from,

    feats<N>.bin   raw float32 [frames][rawDim]              (dataLoader.py:67, write_feats.sh:71)
    keys<N>.txt    "<utterance-id> <nframes>" per line       (dataLoader.py:56-61)
    alis<N>.txt    "<utterance-id> <label> <label> ..."      (dataLoader.py:48-53)

with the same class, constructor and methods (``loadDataFile``, ``loadDataFileDict``,
``loadDataFileAsynch`` / ``getDataAsynch``).  The centre crop of ``imgsize`` columns out of
``rawsize`` (dataLoader.py:63-68) and the (features x frames) orientation of the returned
matrices are kept, so ``data_dict[k]`` feeds ``NNet.costAndGrad`` unchanged.

The one-deep prefetch is a thread instead of the reference's fork + Pipe
(dataLoader.py:22-36): no pickling of a multi-hundred-MB dict through a pipe, and the
loader's NumPy work releases the GIL while the GPU step runs.
"""
import os
import threading

import numpy as np


class DataLoader:
    def __init__(self, filedir_feat, rawsize, imgsize, filedir_ali=None, load_ali=True,
                 load_data=True):
        """
        filedir_feat: directory for feature and key files
        filedir_ali: directory for alignment files. Assumed same as filedir if not given
        """
        self.filedir_feat = filedir_feat
        self.rawsize = rawsize
        self.imgsize = imgsize
        self.filedir_ali = filedir_feat if filedir_ali is None else filedir_ali
        self.load_ali = load_ali
        self.load_data = load_data
        self._thread = None
        self._result = None
        self._error = None

    # -- asynchronous one-deep prefetch (runNNet.py:160,172-175)
    def loadDataFileAsynch(self, filenum):
        self._result, self._error = None, None

        def work():
            try:
                self._result = self.loadDataFileDict(filenum)
            except Exception as e:          # surfaced by getDataAsynch
                self._error = e
        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def getDataAsynch(self):
        assert self._thread is not None, "Error in order of asynch calls."
        self._thread.join()
        self._thread = None
        if self._error is not None:
            raise self._error
        return self._result

    # -- file format
    def loadDataFile(self, filenum):
        keyfile = os.path.join(self.filedir_feat, 'keys%d.txt' % filenum)
        alisfile = os.path.join(self.filedir_ali, 'alis%d.txt' % filenum)
        datafile = os.path.join(self.filedir_feat, 'feats%d.bin' % filenum)
        keys, sizes, data = None, None, None
        alis = {}
        if self.load_ali:
            with open(alisfile, 'r') as fid:
                for line in fid:
                    parts = line.split()
                    if parts:
                        alis[parts[0]] = parts[1:]
        if self.load_data:
            if os.path.exists(keyfile):
                with open(keyfile, 'r') as keyf:
                    uttdat = [u.split() for u in keyf if u.strip()]
                sizes = np.array([np.int32(u[1]) for u in uttdat])
                keys = [u[0] for u in uttdat]
            left = (self.rawsize - self.imgsize) // 2
            right = left + self.imgsize
            data = np.fromfile(datafile, np.float32).reshape(-1, self.rawsize)
            data = data[:np.sum(sizes), left:right]
            return data.T, alis, keys, sizes
        # no data loaded: keys come from the alignments
        return data, alis, list(alis.keys()), sizes

    def loadDataFileDict(self, filenum):
        """Like loadDataFile, but the frames are returned as a dict utterance-id ->
        (imgsize, nframes) float32 matrix."""
        data_mat, alis, keys, sizes = self.loadDataFile(filenum)
        if not self.load_data:
            return None, alis, keys, sizes
        # One copy of the cropped shard, [frames][imgsize] C-order -- into page-locked memory when a
        # GPU is present, so that NNet's host->device staging is a DMA straight out of this buffer
        # -- and per-utterance (imgsize, nframes) Fortran-ordered VIEWS into it (each view keeps the
        # buffer alive).
        frames = data_mat.shape[1]
        shard = _host_buffer(frames, data_mat.shape[0])
        shard[:] = data_mat.T
        data_dict = {}
        start = 0
        for k, s in zip(keys, sizes):
            end = start + int(s)
            data_dict[k] = shard[start:end].T
            start = end
        assert start == frames, "key file and feature file disagree on the frame count"
        return data_dict, alis, keys, sizes


def _host_buffer(rows, cols):
    """float32 [rows][cols]; pinned (page-locked) when torch sees a GPU, plain NumPy otherwise"""
    try:
        import torch
        if torch.cuda.is_available():
            return torch.empty((rows, cols), dtype=torch.float32).pin_memory().numpy()
    except Exception:
        pass
    return np.empty((rows, cols), dtype=np.float32)
