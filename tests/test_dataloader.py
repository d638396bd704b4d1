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


def test_loader_matches_reference_loader(golden):
    """tests/golden/shard/ read by this loader == what the reference's own dataLoader.py:38-95
    (converted with lib2to3 in the build container, tests/golden/make_golden.py) returned for it:
    centre crop and full width, dict and matrix forms, keys, sizes, alignments"""
    import os
    import dataLoader as dl
    from tests.conftest import GOLDEN
    g = golden("loader_ref.npz")
    raw, img = int(g["rawsize"]), int(g["imgsize"])
    for tag, im in (("crop", img), ("full", raw)):
        loader = dl.DataLoader(os.path.join(GOLDEN, "shard") + "/", raw, im)
        data_dict, alis, keys, sizes = loader.loadDataFileDict(1)
        assert list(keys) == [str(k) for k in g[tag + "_keys"]]
        np.testing.assert_array_equal(np.asarray(sizes), g[tag + "_sizes"])
        assert np.asarray(sizes).dtype == g[tag + "_sizes"].dtype
        for i, k in enumerate(keys):
            ref = g["%s_data%d" % (tag, i)]
            assert data_dict[k].shape == ref.shape and data_dict[k].dtype == ref.dtype
            np.testing.assert_array_equal(data_dict[k], ref)
            assert list(alis[k]) == [str(a) for a in g["%s_alis%d" % (tag, i)]]
        mat, _, _, _ = loader.loadDataFile(1)
        np.testing.assert_array_equal(mat, g[tag + "_mat"])
        # no trailing slash works too (the reference concatenates strings and needs it)
        loader2 = dl.DataLoader(os.path.join(GOLDEN, "shard"), raw, im)
        assert list(loader2.loadDataFileDict(1)[2]) == list(keys)


def test_py2_checkpoint_fixture_parses(golden):
    """tests/golden/ref_py2_params.pk is a params.pk in Python-2 cPickle protocol-0 bytes
    (sgd.py:36-42 + brnnet.py:258-267): it needs encoding='latin1' under Python 3"""
    import os
    import pickle
    from tests.conftest import GOLDEN
    g = golden("ref_py2_params.npz")
    with open(os.path.join(GOLDEN, "ref_py2_params.pk"), "rb") as f:
        with pytest.raises(UnicodeDecodeError):
            pickle.load(f)
        f.seek(0)
        it, costt, expcost, vel = pickle.load(f, encoding="latin1")
        par = pickle.load(f, encoding="latin1")
    assert it == int(g["it"])
    np.testing.assert_array_equal(costt, g["costt"])
    np.testing.assert_array_equal(expcost, g["expcost"])
    for i, ((vw, vb), (w, b)) in enumerate(zip(vel, par)):
        for got, key in ((vw, "vw"), (vb, "vb"), (w, "w"), (b, "b")):
            assert got.dtype == np.float32
            np.testing.assert_array_equal(got, g["%s%d" % (key, i)])
