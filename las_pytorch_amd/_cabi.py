"""ctypes binding of liblas_hip.so (include/las_hip.h).  No fallback: if the library is missing the
import of the product path fails loudly — there is no CPU or eager-PyTorch substitute."""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblas_hip.so")
MAX_L = 4

FLAG_STASH = 1
FLAG_FORCE_GENERIC = 2
FLAG_TEACHER_FORCED = 4     # las_speller_bwd: the forward that filled `reserve` was teacher-forced (same flags / error word)
FLAG_GRADS_ZEROED = 16      # las_pblstm_bwd / las_speller_bwd: the gradient block was zeroed by the caller (flat buffer, once per step)
FLAG_DEFER_DW = 32          # ... their weight-gradient GEMM group may still be running on the library's side stream on return (las_join_deferred)
FLAG_GEMM_F32 = 8           # this call's GEMMs on the fp32 matrix pipe (per call; the process-wide default is option GEMM_ARITH)

_f = C.c_void_p   # every device pointer is passed as an integer address


class SpellerDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("Tp", C.c_int), ("D", C.c_int), ("Hs", C.c_int), ("V", C.c_int), ("M", C.c_int),
                ("L", C.c_int), ("use_mlp", C.c_int), ("relu", C.c_int), ("multi_head", C.c_int),
                ("w_ih", _f * MAX_L), ("w_hh", _f * MAX_L), ("b_ih", _f * MAX_L), ("b_hh", _f * MAX_L),
                ("w_phi", _f), ("b_phi", _f), ("w_psi", _f), ("b_psi", _f), ("w_c", _f), ("b_c", _f),
                ("w_dr", _f), ("b_dr", _f)]


class GemmDescC(C.Structure):
    _fields_ = [("A", _f), ("B", _f), ("C", _f), ("A2", _f), ("B2", _f), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("K1", C.c_int),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("a_kc", C.c_int), ("b_kc", C.c_int), ("accumulate", C.c_int),
                ("c_zeroed", C.c_int), ("planes", C.c_int)]


class SpellerGrads(C.Structure):
    _fields_ = [("dw_ih", _f * MAX_L), ("dw_hh", _f * MAX_L), ("db_ih", _f * MAX_L), ("db_hh", _f * MAX_L),
                ("dw_phi", _f), ("db_phi", _f), ("dw_psi", _f), ("db_psi", _f), ("dw_c", _f), ("db_c", _f),
                ("dw_dr", _f), ("db_dr", _f), ("dfeat", _f)]


# name -> (restype, argtypes); every symbol include/las_hip.h declares
PROTOTYPES = {
    "las_abi_version": (C.c_int, []),
    "las_last_error": (C.c_char_p, []),
    "las_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "las_gemm_check": (C.c_int, []),
    "las_join_deferred": (C.c_int, [C.c_void_p]),
    "las_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "las_clip_adam_workspace_floats": (C.c_size_t, []),
    "las_clip_adam": (C.c_int, [C.POINTER(_f), C.POINTER(C.c_int64), C.c_int, _f, _f, _f, C.c_float, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.c_int, _f, _f, _f, _f]),
    "las_debug_persist_trace": (None, [_f]),
    "las_debug_persist_bwd_trace": (None, [_f]),
    "las_debug_big_trace": (None, [_f]),
    "las_debug_big_bwd_trace": (None, [_f]),
    "las_debug_kernel_ms": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "las_debug_xcd_probe": (None, [C.c_void_p]),
    "las_debug_last_path": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "las_pblstm_reserve_floats": (C.c_size_t, [C.c_int] * 4),
    "las_pblstm_fwd": (C.c_int, [_f, C.c_int, C.c_int, C.c_int, C.c_int] + [_f] * 8 + [_f, _f, _f, C.c_int, _f]),
    "las_pblstm_bwd_workspace_floats": (C.c_size_t, [C.c_int] * 3),
    "las_pblstm_bwd": (C.c_int, [_f, _f, C.c_int, C.c_int, C.c_int, C.c_int] + [_f] * 4 + [_f, _f, _f] + [_f] * 8
                       + [_f, C.c_int, _f]),
    "las_attn_keys_fwd": (C.c_int, [C.POINTER(SpellerDesc), _f, _f, _f]),
    "las_speller_reserve_floats": (C.c_size_t, [C.POINTER(SpellerDesc), C.c_int]),
    "las_speller_decode_batch": (C.c_int, [C.POINTER(SpellerDesc), C.c_int, C.c_int]),
    "las_speller_fwd": (C.c_int, [C.POINTER(SpellerDesc), _f, _f, _f, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f, _f, _f, _f,
                                  _f, C.c_int, _f]),
    "las_speller_step_workspace_floats": (C.c_size_t, [C.POINTER(SpellerDesc)]),
    "las_speller_step_reserve_floats": (C.c_size_t, [C.POINTER(SpellerDesc)]),
    "las_speller_step_fwd": (C.c_int, [C.POINTER(SpellerDesc)] + [_f] * 13),
    "las_speller_step_bwd_workspace_floats": (C.c_size_t, [C.POINTER(SpellerDesc)]),
    "las_speller_step_bwd": (C.c_int, [C.POINTER(SpellerDesc)] + [_f] * 18 + [C.POINTER(SpellerGrads), _f, _f]),
    "las_attention_reserve_floats": (C.c_size_t, [C.POINTER(SpellerDesc)]),
    "las_attention_fwd": (C.c_int, [C.POINTER(SpellerDesc)] + [_f] * 7),
    "las_attention_bwd_workspace_floats": (C.c_size_t, [C.POINTER(SpellerDesc)]),
    "las_attention_bwd": (C.c_int, [C.POINTER(SpellerDesc)] + [_f] * 7 + [C.POINTER(SpellerGrads), _f, _f]),
    "las_speller_bwd_workspace_floats": (C.c_size_t, [C.POINTER(SpellerDesc), C.c_int]),
    "las_speller_bwd": (C.c_int, [C.POINTER(SpellerDesc), _f, _f, _f, _f, _f, C.c_int, C.c_int, _f, _f,
                                  C.POINTER(SpellerGrads), _f, C.c_int, _f]),
    "las_ls_loss": (C.c_int, [_f, C.c_int64, C.c_int64, _f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f, _f, C.c_int64,
                              C.c_int64, _f, _f]),
    "las_letter_error_rate": (C.c_int, [_f, C.c_int64, C.c_int64, _f, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f, _f]),
    "las_collate_pad": (C.c_int, [_f, _f, _f, _f] + [C.c_int] * 5 + [_f, _f, _f]),
    "las_comm_uid": (C.c_int, [C.c_char_p]),
    "las_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_char_p]),
    "las_allreduce_f32": (C.c_int, [_f, C.c_size_t, C.c_int, _f]),
    "las_comm_destroy": (C.c_int, []),
    "las_gemm_f32": (C.c_int, [_f] * 5 + [C.c_int] * 3 + [C.c_int64] * 3 + [C.c_int, C.c_int, C.c_int]
                     + [C.c_int64] * 3 + [C.c_int, C.c_int, C.c_int, _f]),
    "las_gemm_f32_group": (C.c_int, [C.POINTER(GemmDescC), C.c_int, _f]),
    "las_planes_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "las_split_planes": (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, _f, C.c_int64, _f]),
    "las_gemm_planes": (C.c_int, [_f] * 5 + [C.c_int] * 3 + [C.c_int64] * 3 + [C.c_int, C.c_int, C.c_int]
                        + [C.c_int64] * 3 + [C.c_int, C.c_int, C.c_int, _f]),
    "las_gemm_get_arith": (C.c_int, []),
    "las_gemm_set_arith": (None, [C.c_int]),
    "las_gemm_set_tuning": (None, [C.c_int, C.c_int64]),
    "las_rec_xbuf_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "las_pblstm_rec_fwd": (C.c_int, [_f] * 6 + [C.c_int] * 3 + [_f, _f, C.c_int, _f]),
}

_lib = None


def lib():
    """Load liblas_hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C las_pytorch_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the LAS hot path has no CPU/eager fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)      # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def set_option(key, value):
    """las_set_option: process-wide run-time switch of the library (see include/las_hip.h for the keys)."""
    check(lib().las_set_option(str(key).encode(), int(value)))


def get_option(key):
    out = C.c_int64(0)
    check(lib().las_get_option(str(key).encode(), C.byref(out)))
    return int(out.value)


PATH_REC_FWD, PATH_REC_BWD, PATH_DECODE_FWD, PATH_DECODE_BWD, PATH_GEMM, PATH_DW = range(6)


def last_path(which):
    """las_debug_last_path: name of the kernel family the most recent call launched for slot ``which`` (see include/las_hip.h)."""
    buf = C.create_string_buffer(64)
    check(lib().las_debug_last_path(int(which), buf, 64))
    return buf.value.decode()


def check(rc):
    if rc != 0:
        msg = lib().las_last_error().decode(errors="replace")
        raise RuntimeError(f"liblas_hip error {rc}: {msg}")


def ptr(t):
    """Device address of a contiguous fp32/int tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("the LAS HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("the LAS HIP path needs contiguous tensors")
    return t.data_ptr()


def ptr_strided(t):
    """Device address of a tensor whose strides the callee is told explicitly (no contiguity requirement)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("the LAS HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


_err_words = {}
_err_snap = {}      # device index -> (pinned host copy, event): the asynchronous snapshot poll_device_errors() looks at
_err_snap_time = {}  # device index -> host time the last snapshot was enqueued
_POLL_PERIOD_S = 0.05


def _dev_index(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


_STEP_READBACK_MAX = 4096      # utterances whose letter error rates ride in the step's one device-to-host copy
_err_bufs = {}                 # device index -> int32 [4 error words | 4 floats (loss, spare) | _STEP_READBACK_MAX floats (LER)]


_spin_log2 = 0                 # 0: the kernels' built-in spin budget (2^18 spins, ~42 ms); see set_handoff_spin_log2


def err_word(device):
    """Per-device uint32 the kernels write a nonzero code into when a bounded hand-off spin expires (word 1 behind it: the optional
    extended spin budget, include/las_hip.h).  It is the head of a small buffer whose tail takes a training step's loss and letter error
    rates (``step_readback``), so that the solver reads all three with one copy."""
    key = _dev_index(device)
    w = _err_words.get(key)
    if w is None:
        buf = torch.zeros(8 + _STEP_READBACK_MAX, dtype=torch.int32, device=f"cuda:{key}")
        _err_bufs[key] = buf
        w = buf[:4]
        _err_words[key] = w
        if _spin_log2:
            w[1] = 1 << _spin_log2
    return w


def set_handoff_spin_log2(n):
    """Spin budget of the persistent kernels' bounded hand-off waits: ``2**n`` spins of ~160 ns (0 / <= 18: the built-in 2^18, ~42 ms, sized
    for callers that can re-run a step — ``solver.batch_iterator`` with ``FusedClipAdam`` or in validation).  A caller that cannot roll a
    step back (plain torch optimizer, foreign gradient exchange) asks for 21 (~340 ms, the margin of rounds 1-4): a shared GPU, a debugger
    or a pre-empted queue then has eight times longer before the step is declared lost.  Also read from LAS_HANDOFF_SPIN_LOG2 at import."""
    global _spin_log2
    n = int(n)
    if n and not 18 < n < 31:
        n = 0
    _spin_log2 = n
    for w in _err_words.values():
        w[1] = (1 << n) if n else 0


def handoff_spin_log2():
    return _spin_log2 or 18


def _clear_error(w):
    w[:1].zero_()              # word 1 (the spin budget) stays


def step_readback(device, B):
    """``(loss_out, ler_out)``: a 1-element and a B-element fp32 view behind the device's error word, for ``las_ls_loss`` / ``las_letter_error_rate``
    to write into; ``read_step`` then fetches error word, loss and rates in ONE device-to-host copy.  None when B does not fit."""
    if B > _STEP_READBACK_MAX:
        return None
    err_word(device)
    f = _err_bufs[_dev_index(device)][4:].view(torch.float32)
    return f[0:1], f[4:4 + B]


def read_step(device, B):
    """The step's one synchronisation point: ``(error word, loss (0-d float32 ndarray), [B letter error rates])``."""
    key = _dev_index(device)
    host = _err_bufs[key][:8 + B].cpu()
    f = host[4:].view(torch.float32).numpy()
    return int(host[0]), np.array(f[0], dtype=np.float32), f[4:4 + B].tolist()      # the loss as a 0-d ndarray, like loss.cpu().data.numpy()


class DeviceHandoffError(RuntimeError):
    """A persistent kernel reported a hand-off timeout through the device error word (the call's results are invalid)."""


_defer_polls = 0


class polls_deferred:
    """``with polls_deferred():`` — the non-blocking polls at the top of every Listener / Speller forward do not raise inside
    the block.  ``solver.batch_iterator`` uses it: it checks the error word itself at the step's synchronisation point, where
    all ranks of a data-parallel group reach the same verdict and the step can be re-run as a whole."""

    def __enter__(self):
        global _defer_polls
        _defer_polls += 1

    def __exit__(self, *exc):
        global _defer_polls
        _defer_polls -= 1
        return False


def _raise_device_error(key, v):
    raise DeviceHandoffError(f"liblas_hip device-side failure 0x{v & 0xffffffff:08x} on cuda:{key} "
                       "(an inter-workgroup hand-off of a persistent kernel timed out: another kernel was resident on the "
                       "GPU, or fewer compute units were available than the launch assumed; the results of that call are "
                       "invalid).  The error word has been cleared; LAS_FLAG_FORCE_GENERIC / force_generic selects the "
                       "per-step kernels, which need no co-residency.")


def check_device_errors(words=None):
    """Synchronising check of the device error words: raises if any persistent kernel reported a hand-off timeout and
    clears the word so later launches run normally.  ``solver.batch_iterator`` calls this right after the loss reaches
    the host (the step's existing synchronisation point); tests, smoke and bench call it after their timed regions.
    ``words`` ({device index: value}): error words the caller has just read together with its results (``read_step``: the stream is
    drained, no second copy is made for those devices)."""
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        if words is None:
            torch.cuda.synchronize()
        if lib().las_gemm_check() != 0:   # a stream-K fix-up wait that timed out (host-visible word, needs no device copy)
            raise DeviceHandoffError("liblas_hip: " + lib().las_last_error().decode(errors="replace") +
                                     " (GEMM_SK_FIXUP: the tile that wait belonged to is wrong; the step must be re-run)")
    for key, w in _err_words.items():
        v = int(words[key]) if (words is not None and key in words) else int(w[0].item())
        if v != 0:
            _clear_error(w)
            _err_snap.pop(key, None)
            _raise_device_error(key, v)


def poll_device_errors(device):
    """Non-blocking form used at the top of every ``Listener`` / ``Speller`` forward: looks at the snapshot of the
    error word copied to pinned host memory behind the PREVIOUS call's kernels (raising if it is nonzero) and
    enqueues the next snapshot.  Detection is one call late but costs no host synchronisation."""
    key = _dev_index(device)
    w = _err_words.get(key)
    if w is None or _defer_polls or torch.cuda.is_current_stream_capturing():
        return
    snap = _err_snap.get(key)
    if snap is not None:
        host, ev = snap
        if not ev.query():
            return                        # the previous snapshot has not landed yet: look again next call
        v = int(host[0])
        if v != 0:
            _clear_error(w)
            _err_snap.pop(key, None)
            _raise_device_error(key, v)
        # a new snapshot at most every _POLL_PERIOD_S: each one is a device-to-host copy on the compute stream (4 us + a 6 us bubble in
        # the step's timeline; four forwards per step would otherwise enqueue one almost every step).  Detection through THIS path is
        # therefore up to that much later; solver.batch_iterator reads the word itself at every step's synchronisation point.
        now = time.monotonic()
        if now - _err_snap_time.get(key, 0.0) < _POLL_PERIOD_S:
            return
    else:
        host = torch.zeros(4, dtype=torch.int32).pin_memory()
        ev = torch.cuda.Event()
        _err_snap[key] = (host, ev)
    host, ev = _err_snap[key]
    _err_snap_time[key] = time.monotonic()
    with torch.cuda.device(key):
        host.copy_(w, non_blocking=True)
        ev.record()


try:
    set_handoff_spin_log2(int(os.environ.get("LAS_HANDOFF_SPIN_LOG2", "0")))
except ValueError:
    pass
