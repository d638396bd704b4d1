"""dataLoader.py: on-disk format of the reference's training shards
(ctc_fast/dataLoader.py:38-95, util/swbd/write_feats.sh:71).  CPU only."""
import numpy as np
import pytest


def write_shard(d, n, utts, raw, rs):
    feats = []
    with open(d / ("keys%d.txt" % n), "w") as kf, open(d / ("alis%d.txt" % n), "w") as af:
        for name, T, labels in utts:
            kf.write("%s %d\n" % (name, T))
            af.write("%s %s\n" % (name, " ".join(str(l) for l in labels)))
            feats.append(rs.randn(T, raw).astype(np.float32))
    allf = np.concatenate(feats, axis=0)
    # Kaldi's export may append frames beyond sum(sizes); the loader must ignore them
    np.concatenate([allf, rs.randn(3, raw).astype(np.float32)]).tofile(str(d / ("feats%d.bin" % n)))
    return feats


def test_load_data_file_dict(tmp_path):
    import dataLoader as dl
    rs = np.random.RandomState(0)
    raw, img = 15, 9
    utts = [("utt_a", 7, [3, 1, 4]), ("utt_b", 12, [1, 5]), ("utt_c", 1, [9])]
    feats = write_shard(tmp_path, 1, utts, raw, rs)
    loader = dl.DataLoader(str(tmp_path) + "/", raw, img)
    data_dict, alis, keys, sizes = loader.loadDataFileDict(1)
    assert keys == ["utt_a", "utt_b", "utt_c"]
    np.testing.assert_array_equal(sizes, [7, 12, 1])
    left = (raw - img) // 2
    for (name, T, labels), f in zip(utts, feats):
        x = data_dict[name]
        assert x.shape == (img, T) and x.dtype == np.float32
        np.testing.assert_array_equal(x, f[:, left:left + img].T)       # centre crop, (dim, frames)
        assert alis[name] == [str(l) for l in labels]
        assert np.array(alis[name], dtype=np.int32).tolist() == labels  # sgd.py:82
    mat, _, _, _ = loader.loadDataFile(1)
    assert mat.shape == (img, 20)


def test_async_prefetch_and_alignment_only(tmp_path):
    import dataLoader as dl
    rs = np.random.RandomState(1)
    write_shard(tmp_path, 1, [("a", 4, [1])], 6, rs)
    write_shard(tmp_path, 2, [("b", 5, [2, 2])], 6, rs)
    loader = dl.DataLoader(str(tmp_path) + "/", 6, 6)
    with pytest.raises(AssertionError):
        loader.getDataAsynch()                       # "Error in order of asynch calls."
    loader.loadDataFileAsynch(2)
    data_dict, alis, keys, sizes = loader.getDataAsynch()
    assert keys == ["b"] and data_dict["b"].shape == (6, 5)
    loader.loadDataFileAsynch(7)                     # missing shard: error surfaces on get
    with pytest.raises(FileNotFoundError):
        loader.getDataAsynch()
    only_ali = dl.DataLoader(str(tmp_path) + "/", 6, 6, load_data=False)
    data, alis, keys, sizes = only_ali.loadDataFileDict(1)
    assert data is None and list(keys) == ["a"] and alis["a"] == ["1"]
