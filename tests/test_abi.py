"""CPU-side checks of the drop-in boundary: libsctc_hip.so loads here (no GPU) and
exports exactly the functions include/sctc.h declares; the ctypes struct mirrors
match the C structs; the host-side mirror exposes the reference's names and fails
loudly (no fallback) without a GPU.  No compute calls."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sctc.h")


@pytest.fixture(scope="module")
def sctc():
    import __graft_entry__ as ge
    import _sctc
    if not os.path.exists(_sctc.LIB_PATH):
        ge.build()
    return _sctc


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sctc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(sctc):
    L = sctc.lib()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libsctc_hip.so does not export %s" % n
    assert sorted(sctc.PROTOTYPES) == names, "ctypes prototypes out of sync with include/sctc.h"
    assert L.sctc_abi_version() == 6        # v6: sctc_brnn_ctc_workspace_bytes / _set_ctc_workspace; v5: sctc_brnn_allreduce_grads (v4: sctc_device_pci_bus_id; v3: shared-device mode, recurrent_path)


def test_struct_mirrors_match_the_header(sctc, tmp_path):
    """compile a tiny C program against include/sctc.h and compare sizeof/offsetof"""
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "sctc.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sctc_ctc_batch),'
        ' offsetof(sctc_ctc_batch, rowbase_dev), sizeof(sctc_brnn_config),'
        ' offsetof(sctc_brnn_config, operand_dtype), sizeof(sctc_tensor_info), sizeof(sctc_brnn_sizes),'
        ' sizeof(sctc_minibatch), offsetof(sctc_minibatch, U_b)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(sctc.CtcBatch), sctc.CtcBatch.rowbase_dev.offset,
            ctypes.sizeof(sctc.BrnnConfig), sctc.BrnnConfig.operand_dtype.offset,
            ctypes.sizeof(sctc.TensorInfo), ctypes.sizeof(sctc.BrnnSizes),
            ctypes.sizeof(sctc.Minibatch), sctc.Minibatch.U_b.offset]
    assert got == want


def test_argument_errors_need_no_gpu(sctc):
    L = sctc.lib()
    cfg = sctc.BrnnConfig(20, 6, 30, 3, 2, 10, 1, 20.0, 0.0, 1)
    sizes = sctc.BrnnSizes()
    assert L.sctc_brnn_query(ctypes.byref(cfg), ctypes.byref(sizes)) == 0
    # weights [32 rows][64-float stride], biases 32 each, Wf, Wb
    assert sizes.n_tensors == 2 * 4 + 2
    assert sizes.param_count == 20 * 30 + 30 + 2 * (30 * 30 + 30) + 30 * 6 + 6 + 2 * 30 * 30
    assert sizes.param_elems == 4 * (32 * 64 + 32) + 2 * 32 * 64
    assert sizes.workspace_bytes > 0
    big = sctc.BrnnConfig(20, 6, 4096, 3, 2, 10, 1, 20.0, 0.0, 1)   # 512 workgroups per pass: cannot be
    assert L.sctc_brnn_query(ctypes.byref(big), ctypes.byref(sizes)) == 0   # persistent -> per-step launches
    wide = sctc.BrnnConfig(20, 300, 30, 3, 2, 10, 1, 20.0, 0.0, 1)   # alphabets beyond 256 symbols: accepted since
    assert L.sctc_brnn_query(ctypes.byref(wide), ctypes.byref(sizes)) == 0   # round 5 (ctc_generic.hip), like the reference
    bad = sctc.BrnnConfig(20, 1, 30, 3, 2, 10, 1, 20.0, 0.0, 1)
    assert L.sctc_brnn_query(ctypes.byref(bad), ctypes.byref(sizes)) == -1
    assert b"dimensions" in L.sctc_last_error()
    # the C-level collective entry rejects a missing handle / communicator before anything is touched
    assert L.sctc_brnn_allreduce_grads(None, None, None, None, None, 0, 1) == -1
    with pytest.raises(ValueError):
        sctc.check(-1, "x")
    T = np.array([5], dtype=np.int32)
    U = np.array([0], dtype=np.int32)
    off = np.zeros(1, dtype=np.int64)
    lab = np.zeros(1, dtype=np.int32)
    bt = sctc.CtcBatch(1, 4, 0, sctc.F64, 4, sctc.i32(T), sctc.i32(U), sctc.i64(off),
                       sctc.i32(lab), sctc.i64(off), None)
    assert L.sctc_ctc_workspace_bytes(ctypes.byref(bt)) == 0      # empty label sequence rejected
    U[0] = 2
    n = L.sctc_ctc_workspace_bytes(ctypes.byref(bt))
    assert n >= 5 * 6 * 8                                         # fused path: ONE packed float64 half lattice per direction
    os.environ["SCTC_CTC_FUSED"] = "0"
    try:
        assert L.sctc_ctc_workspace_bytes(ctypes.byref(bt)) >= 2 * 5 * 128 * 8   # two float64 lattices of 5 x 128
    finally:
        del os.environ["SCTC_CTC_FUSED"]
    U[0] = 1500                                                   # 2U+1 > 2048: the generic kernels, no rejection
    T[0] = 1600
    assert L.sctc_ctc_workspace_bytes(ctypes.byref(bt)) >= 2 * 1600 * 3001 * 8
    # rows of 1025..2048 states (round 6): from 18 utterances on the wide fused kernel -- ONE packed row store, 32-bit on
    # float32 probabilities -- below that (and with SCTC_CTC_WIDE=0) two float64 lattices of 2048-state rows
    def need(B, dtype, T=1000, Ul=800):
        Tb, Ub = np.full(B, T, dtype=np.int32), np.full(B, Ul, dtype=np.int32)
        o, lb = np.arange(B, dtype=np.int64) * T, np.zeros(B * Ul, dtype=np.int32)
        lo = np.arange(B, dtype=np.int64) * Ul
        b = sctc.CtcBatch(B, 33, 0, dtype, 33, sctc.i32(Tb), sctc.i32(Ub), sctc.i64(o), sctc.i32(lb), sctc.i64(lo), None)
        return L.sctc_ctc_workspace_bytes(ctypes.byref(b))
    assert need(17, sctc.F32) >= 2 * 17 * 1000 * 2048 * 8
    assert 18 * 1000 * 1604 * 4 <= need(18, sctc.F32) <= 18 * 1000 * 1604 * 4 + 18 * 40000
    assert 18 * 1000 * 1604 * 8 <= need(18, sctc.F64) <= 18 * 1000 * 1604 * 8 + 18 * 40000
    for env, lo_bound in (("SCTC_CTC_WIDE", "0"), ("SCTC_CTC_FUSED", "0")):
        os.environ[env] = lo_bound
        try:
            assert need(18, sctc.F32) >= 2 * 18 * 1000 * 2048 * 8
        finally:
            del os.environ[env]
    os.environ["SCTC_CTC_WIDE_MIN_B"] = "1"
    try:
        assert need(1, sctc.F32) <= 1000 * 1604 * 4 + 40000
    finally:
        del os.environ["SCTC_CTC_WIDE_MIN_B"]


def test_reference_surface_names(sctc):
    import ctc_fast
    from nnets import brnnet
    import cudamat as cm
    assert callable(ctc_fast.ctc_loss) and callable(ctc_fast.decode_best_path)
    # ctc_fast.pyx:6 import side effect
    assert np.geterr()["divide"] == "raise" and np.geterr()["invalid"] == "raise"
    for name in ("initParams", "paramCount", "setViews", "costAndGrad", "updateParams", "toFile",
                 "fromFile", "check_grad"):
        assert callable(getattr(brnnet.NNet, name))
    for name in ("mult", "add_mult", "euclid_norm", "copy_to_host", "copy_to_device"):
        assert callable(getattr(cm.CUDAMatrix, name))
    assert callable(cm.cuda_set_device) and callable(cm.cublas_init)
    assert cm.padded_layout(1824, 483) == (1824, 512) and cm.padded_layout(33, 1) == (64, 1)
    assert cm.padded_layout(1824, 1824) == (1824, 1856)
    assert ctc_fast.collapse_best_path(np.array([0, 3, 3, 0, 3, 1, 4, 4, 8, 5, 0, 5])) == \
        ([3, 3, 4, 5, 5], [2, 4, 7, 9, 11])


def test_no_cpu_fallback(sctc):
    """without a GPU every compute entry point raises instead of computing on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ctc_fast
    from nnets import brnnet
    y = np.asfortranarray(np.full((3, 4), 1.0 / 3))
    with pytest.raises(sctc.SctcError):
        ctc_fast.ctc_loss(y, np.array([1], dtype=np.int32))
    # argument checking happens before the device is touched, like the Cython signature
    with pytest.raises(ValueError):
        ctc_fast.ctc_loss(np.ascontiguousarray(np.full((3, 4), 1.0 / 3)), np.array([1], dtype=np.int32))
    net = brnnet.NNet(5, 4, 32, 2, 10, temporalLayer=1)
    with pytest.raises(sctc.SctcError):
        net.initParams()
