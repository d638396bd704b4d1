"""ctypes binding of libsctc_hip.so (include/sctc.h) -- the only way the host-side
mirror reaches the GPU.  There is NO CPU fallback: if the library is missing this
module raises, and every compute entry point raises when no HIP device is present.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCTC_LIB_PATH") or os.path.join(_HERE, "libsctc_hip.so")   # override: kernel-variant experiments

F32, F64, F16, BF16, BF16X3 = 0, 1, 2, 3, 4

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p


class CtcBatch(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("A", ctypes.c_int32), ("blank", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("ld", ctypes.c_int64), ("T_b", c_i32p),
                ("U_b", c_i32p), ("frame_off", c_i64p), ("labels", c_i32p),
                ("label_off", c_i64p), ("rowbase_dev", vp)]


class BrnnConfig(ctypes.Structure):
    _fields_ = [("input_dim", ctypes.c_int32), ("output_dim", ctypes.c_int32),
                ("layer_size", ctypes.c_int32), ("num_layers", ctypes.c_int32),
                ("temporal_layer", ctypes.c_int32), ("max_frames", ctypes.c_int32),
                ("max_utts", ctypes.c_int32), ("max_act", ctypes.c_float),
                ("reg", ctypes.c_float), ("train", ctypes.c_int32),
                ("operand_dtype", ctypes.c_int32)]


class TensorInfo(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_int64), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("ld", ctypes.c_int32), ("kind", ctypes.c_int32)]


class BrnnSizes(ctypes.Structure):
    _fields_ = [("param_elems", ctypes.c_int64), ("param_count", ctypes.c_int64),
                ("workspace_bytes", ctypes.c_size_t), ("n_tensors", ctypes.c_int32)]


class Minibatch(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("T_b", c_i32p), ("feats_dev", vp), ("labels", c_i32p),
                ("U_b", c_i32p)]


N_PHASES = 6

# name -> (restype, argtypes); every symbol include/sctc.h declares
PROTOTYPES = {
    "sctc_abi_version": (ctypes.c_int, []),
    "sctc_last_error": (ctypes.c_char_p, []),
    "sctc_set_device": (ctypes.c_int, [ctypes.c_int]),
    "sctc_device_info": (ctypes.c_int, [c_i32p, c_i32p, c_i64p, ctypes.c_char_p, ctypes.c_int]),
    "sctc_set_shared_device": (ctypes.c_int, [ctypes.c_int32]),
    "sctc_shared_device": (ctypes.c_int, []),
    "sctc_device_pci_bus_id": (ctypes.c_int, [ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]),
    "sctc_ctc_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(CtcBatch)]),
    "sctc_ctc_loss_batch": (ctypes.c_int, [ctypes.POINTER(CtcBatch), vp, vp, vp, vp, vp,
                                           ctypes.c_size_t, vp]),
    "sctc_softmax_rows": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64,
                                         vp]),
    "sctc_argmax_rows": (ctypes.c_int, [vp, ctypes.c_int32, vp, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_int64, vp]),
    "sctc_brnn_query": (ctypes.c_int, [ctypes.POINTER(BrnnConfig), ctypes.POINTER(BrnnSizes)]),
    "sctc_brnn_create": (ctypes.c_int, [ctypes.POINTER(BrnnConfig), vp, vp, vp, ctypes.c_size_t,
                                        ctypes.POINTER(vp)]),
    "sctc_brnn_destroy": (ctypes.c_int, [vp]),
    "sctc_brnn_tensor_info": (ctypes.c_int, [vp, ctypes.c_int32, ctypes.POINTER(TensorInfo)]),
    "sctc_brnn_cost_and_grad": (ctypes.c_int, [vp, ctypes.POINTER(Minibatch), ctypes.c_int32,
                                               c_f64p, c_i32p, c_f64p, vp]),
    "sctc_brnn_cost_and_grad_async": (ctypes.c_int, [vp, ctypes.POINTER(Minibatch),
                                                     ctypes.c_int32, vp, vp, vp]),
    "sctc_brnn_check": (ctypes.c_int, [vp, vp]),
    "sctc_brnn_recurrent_path": (ctypes.c_int, [vp, c_i32p, c_i32p, c_i32p]),
    "sctc_brnn_grad_event": (vp, [vp, ctypes.c_int32]),
    "sctc_stream_wait_event": (ctypes.c_int, [vp, vp]),
    "sctc_brnn_forward": (ctypes.c_int, [vp, ctypes.POINTER(Minibatch), vp, vp]),
    "sctc_brnn_ctc_workspace_bytes": (ctypes.c_int, [vp, ctypes.POINTER(Minibatch), ctypes.POINTER(ctypes.c_size_t),
                                                     ctypes.POINTER(ctypes.c_size_t)]),
    "sctc_brnn_set_ctc_workspace": (ctypes.c_int, [vp, vp, ctypes.c_size_t]),
    "sctc_brnn_set_profiling": (ctypes.c_int, [vp, ctypes.c_int32]),
    "sctc_brnn_debug_read": (ctypes.c_int, [vp, vp, ctypes.c_int32]),
    "sctc_brnn_debug_buffer": (ctypes.c_int, [vp, ctypes.c_int32, ctypes.POINTER(vp), c_i64p, c_i64p, c_i64p]),
    "sctc_brnn_phase_ms": (ctypes.c_int, [vp, c_f32p]),
    "sctc_brnn_flops": (ctypes.c_int, [vp, ctypes.POINTER(Minibatch), c_f64p, c_f64p, c_f64p]),
    "sctc_gemm_f32": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int32, vp, ctypes.c_int64,
                                     ctypes.c_int32, vp, ctypes.c_int64, ctypes.c_int32,
                                     ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int32, vp,
                                     ctypes.c_size_t, vp]),
    "sctc_gemm_h16": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int32, vp, ctypes.c_int64,
                                     ctypes.c_int32, vp, ctypes.c_int64, ctypes.c_int32,
                                     ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int32,
                                     ctypes.c_int32, vp, ctypes.c_size_t, vp]),
    "sctc_axpy": (ctypes.c_int, [vp, vp, ctypes.c_float, ctypes.c_int64, vp]),
    "sctc_scale": (ctypes.c_int, [vp, ctypes.c_float, ctypes.c_int64, vp]),
    "sctc_sumsq": (ctypes.c_int, [vp, ctypes.c_int64, vp, vp, ctypes.c_size_t, vp]),
    "sctc_nesterov_step": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_float, ctypes.c_float, vp,
                                          vp]),
    "sctc_brnn_allreduce_grads": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32]),
    "sctc_sumsq_reg": (ctypes.c_int, [vp, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int64,
                                      c_i64p, ctypes.c_int32, vp, vp, ctypes.c_size_t, vp]),
    "sctc_nesterov_step_reg": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_float, c_i64p, ctypes.c_int32, vp, vp]),
}

FLAG_SYNC_SKIP = 1
FLAG_ACCUMULATE = 2
FLAG_NO_REG_GRAD = 4

_lib = None


class SctcError(RuntimeError):
    pass


def device_bus_id(device=-1):
    """PCI bus id of HIP device `device` (-1: current): the physical identity of the GPU"""
    buf = ctypes.create_string_buffer(64)
    check(lib().sctc_device_pci_bus_id(device, buf, 64), "device_pci_bus_id")
    return buf.value.decode()


def shared_device_from_ids(ids, rank):
    """ids[r] = (hostname, bus id) of rank r.  True iff another rank sits on this rank's GPU."""
    me = tuple(ids[rank])
    return sum(1 for x in ids if tuple(x) == me) > 1


def resolve_shared_device(group=None, my_id=None, log=True, device=None):
    """Shared-device mode decided by PHYSICAL device identity: every rank
    contributes (hostname, PCI bus id of its device) to an all-gather over the process group; the
    mode is switched ON iff two ranks report the same pair, whatever device ordinals, visibility
    masks or LOCAL_WORLD_SIZE say.  An explicit SCTC_SHARED_DEVICE in the environment wins.
    `device`: the HIP device ordinal the rank's model lives on (dist_sgd.DataParallel passes the device of
    the net's gradient buffer); None = the current device -- which is device 0 on every rank that has
    not called torch.cuda.set_device yet, and then every rank would report the same GPU.
    This function only ever RAISES the mode: a mode that a timeout or the per-device activity markers of
    a foreign process switched on earlier stays on.  NOTE: it is a COLLECTIVE over `group` (an
    all_gather_object): every rank must call it, at the same point -- DataParallel() does, in its
    constructor.  Returns (shared, ids).  One line per rank on stderr."""
    import socket
    import sys
    import torch.distributed as dist
    L = lib()
    if my_id is None:
        my_id = (socket.gethostname(), device_bus_id(-1 if device is None else int(device)))
    rank, ids = 0, [tuple(my_id)]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        rank = dist.get_rank(group)
        gathered = [None] * dist.get_world_size(group)
        dist.all_gather_object(gathered, tuple(my_id), group=group)
        ids = [tuple(x) for x in gathered]
    if "SCTC_SHARED_DEVICE" in os.environ:
        shared, why = bool(L.sctc_shared_device()), "SCTC_SHARED_DEVICE in the environment"
    else:
        shared, why = shared_device_from_ids(ids, rank), "bus ids of %d rank(s)" % len(ids)
        if shared:
            L.sctc_set_shared_device(1)
        elif L.sctc_shared_device():
            why += "; the mode was already on (a timeout or another process on this GPU) and stays on"
            shared = True
    if log:
        sys.stderr.write("sctc: rank %d -> %s %s, shared=%d (%s)\n"
                         % (rank, my_id[0], my_id[1], int(shared), why))
        sys.stderr.flush()
    return shared, ids


def lib():
    """Loads libsctc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libsctc_hip.so is missing (%s): build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` -- there is no CPU fallback" % LIB_PATH)
        # torch ships its own libamdhip64; load it FIRST so that this library binds to the same HIP
        # runtime instance (two runtimes in one process: "no ROCm-capable device is detected")
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if L.sctc_abi_version() != 6:
            raise ImportError("libsctc_hip.so ABI version mismatch")
        _lib = L
        # shared-device mode is not guessed here: the library finds out by itself how many processes use
        # the physical GPU (marker files keyed by PCI bus id, csrc/recurrent.hip), and ranks of a process
        # group exchange their bus ids (resolve_shared_device, called by dist_sgd.DataParallel)
    return _lib


def check(rc, what=""):
    if rc < 0:
        msg = lib().sctc_last_error()
        msg = msg.decode() if msg else ""
        if rc == -1:
            raise ValueError("%s: %s" % (what, msg))
        raise SctcError("%s failed (rc=%d): %s" % (what, rc, msg))
    return rc


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise SctcError("no HIP device visible: the stanford-ctc MI355X path has no CPU fallback")
    return torch


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def i32(arr):
    return arr.ctypes.data_as(c_i32p)


def i64(arr):
    return arr.ctypes.data_as(c_i64p)
