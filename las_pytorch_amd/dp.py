"""Data-parallel gradient exchange for the LAS hot path: one process per GPU, parameters replicated,
utterances of the minibatch sharded by rank, and ONE all-reduce of a flat fp32 gradient buffer per step
(RCCL over xGMI when the backend is "nccl").  Replaces the reference's ``nn.DataParallel`` wrapper
(train.py:76-78: per-step parameter broadcast + output gather + gradient reduce to GPU 0 in GIL-bound threads).

The batch-coupled quantities of the reference step are (solver/solver.py:43,96; las_model.py:189):
  * the batch-mean loss  -> average of equal-shard rank gradients == full-batch gradient,
  * the global grad-norm clip -> applied AFTER the all-reduce, identically on every rank,
  * the single teacher-forcing coin flip -> ``sync_coin`` / ``restore_coin``: every rank draws it from rank 0's stream.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


class FlatGradAllReducer:
    """Makes every ``p.grad`` a view into one contiguous fp32 buffer so the whole gradient crosses the fabric as a
    single collective (message: S 7.96 MB, P 39.9 MB)."""

    def __init__(self, module: torch.nn.Module, force: bool = False, direct: bool = False, comm=None, defer_dw: bool = False):
        self.force = force          # all-reduce even in a 1-rank group (exercises the collective path)
        self.comm = comm            # optional CabiComm: the collective goes through liblas_hip's las_allreduce_f32
        self.direct = direct
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # 4 extra floats behind the gradients: [n] is the step's device-error flag, which rides in the same collective so that
        # every rank learns whether ANY rank's persistent kernels timed out (solver.batch_iterator then re-runs the step on all)
        self.flat_ext = torch.zeros(n + 4, dtype=torch.float32, device=dev)
        self.flat = self.flat_ext[:n]
        self.flag = self.flat_ext[n:n + 1]
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.numel = n
        # direct=True: the HIP backward kernels write each parameter gradient straight into its view of the flat buffer
        # (no per-parameter AccumulateGrad add kernel).  Requires one use of every parameter per backward and one
        # backward per step — what solver/solver.py:95-97 does.  The tag is per parameter (it names THIS buffer), so
        # other modules in the process keep ordinary autograd accumulation; see las_model._direct_targets.
        base = self.flat.untyped_storage().data_ptr()
        self.zero_epoch = 0         # zero() calls so far: a backward kernel may skip its own fill of a gradient block that was not written since
        for p in self.params:
            if direct:
                p._las_direct_base = base
                p._las_direct_owner = self
                p._las_written_epoch = -1
            elif hasattr(p, "_las_direct_base"):
                del p._las_direct_base
        module._las_flat_reducer = self      # solver.batch_iterator picks it up (zero / all-reduce / clip on the flat buffer)
        # Deferred weight gradients (LAS_FLAG_DEFER_DW, direct writes only): inside ``with reducer.deferring():`` the backward entry points may
        # leave their weight-gradient GEMM groups running on the library's side stream — hidden under the next layer's recurrence where the
        # batch leaves XCDs free (B <= 8 at paper size) — and ``join_deferred()`` (called by every consumer of the flat buffer here, and at
        # the end of the block) makes the current stream wait for them.  The tensors those GEMMs read are kept alive until then.
        # OFF by default (``defer_dw`` / env LAS_DEFER_DW=1): measured at BASELINE configs[4] (B = 8, T = 3000) the step gains 0.0 - 0.09 ms of
        # the 0.41 ms the hidden GEMMs take alone — a GEMM group beside a resident recurrence runs ~9x slower than alone even on disjoint
        # XCDs and slows the critical-path launches around it (profiles/r06_defer_dw.txt).
        import os
        self.defer_dw = bool(defer_dw) or os.environ.get("LAS_DEFER_DW") == "1"
        self._defer_active = False
        self._deferred_keep = []

    def deferring(self):
        """Context manager around forward + backward of one step: see ``_defer_active`` above.  Only callers that reach the gradients through
        this object (``allreduce_mean`` / ``clip_`` / ``FusedClipAdam``) or after the block may use it."""
        red = self

        class _Ctx:
            def __enter__(self_inner):
                red._defer_active = bool(red.defer_dw and red.direct and red.flat.is_cuda)
                return red

            def __exit__(self_inner, *exc):
                red._defer_active = False
                red.join_deferred()
                return False
        return _Ctx()

    def join_deferred(self):
        """The current stream waits for the deferred weight-gradient work of this step (no-op when there is none)."""
        if self._deferred_keep:
            from . import _cabi
            with torch.cuda.device(self.flat.device):
                _cabi.check(_cabi.lib().las_join_deferred(_cabi.stream_ptr()))
            self._deferred_keep.clear()

    def zero(self):
        """Use instead of ``optimizer.zero_grad()`` (which would drop the views with set_to_none=True).

        HIP-graph note: the backward entry points skip their own fill of a gradient block this call zeroed (``LAS_FLAG_GRADS_ZEROED``), a
        host-side decision that a stream capture bakes into the graph.  The flag is therefore claimed during a capture only when this
        ``zero()`` was recorded in the same capture (``zero_in_capture``) — every replay then zeroes before it skips."""
        self.join_deferred()
        self.flat_ext.zero_()
        self.zero_epoch += 1
        self.zero_in_capture = bool(self.flat.is_cuda and torch.cuda.is_current_stream_capturing())

    def _collective(self):
        """True when allreduce_mean() really exchanges data (more than one rank, or ``force``)."""
        if self.comm is not None:
            return self.comm.world > 1 or self.force
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force)

    def error_flag_ptr(self):
        """What the fused optimizer's device-side guard looks at: the all-reduced flag when the gradients were exchanged (any
        rank's timeout invalidates every rank's averaged gradient), this device's error word otherwise.  None on CPU."""
        if not self.flat.is_cuda:
            return None
        if self._collective():
            return self.flag.data_ptr()
        from . import _cabi
        return _cabi.err_word(self.flat.device).data_ptr()

    def inject_error(self, code=-1.0):
        """Tests / hosts without the device error word: put ``code`` into this rank's flag element; the next ``allreduce_mean``
        carries it to every rank (on a GPU the flag is overwritten with the device error word right before the collective)."""
        self.flag.fill_(float(code))

    def check_views(self):
        """Every ``p.grad`` must still be its view of the flat buffer: ``optimizer.zero_grad()`` with torch's default
        ``set_to_none=True`` silently replaces them, after which the collective would reduce a buffer of zeros while
        the optimizer steps on unsynchronised local gradients.  Fail loudly instead."""
        self.join_deferred()              # every consumer of the flat gradient comes through here first
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                raise RuntimeError("FlatGradAllReducer: a parameter's .grad no longer aliases the flat gradient buffer "
                                   "(use reducer.zero() or optimizer.zero_grad(set_to_none=False), not zero_grad())")

    def allreduce_mean(self):
        """Average the flat gradient over the ranks with ONE collective (RCCL ``ncclAvg`` when the backend is nccl)."""
        self.check_views()
        if not self._collective():
            return
        if self.flat.is_cuda:       # this rank's device error word -> the flag element (codes are negative int32: averages cannot cancel)
            from . import _cabi
            self.flag.copy_(_cabi.err_word(self.flat.device)[:1])
        if self.comm is not None:
            self.comm.allreduce_(self.flat_ext, average=True)
        elif dist.get_backend() == "nccl":
            dist.all_reduce(self.flat_ext, op=dist.ReduceOp.AVG)
        else:                                   # gloo has no AVG
            dist.all_reduce(self.flat_ext, op=dist.ReduceOp.SUM)
            self.flat_ext.div_(dist.get_world_size())

    def clip_(self, max_norm=1.0):
        """clip_grad_norm_(params, max_norm) on the flat buffer (solver/solver.py:96), same formula as torch's."""
        self.join_deferred()
        total = torch.linalg.vector_norm(self.flat)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.flat.mul_(coef)
        return total


class CabiComm:
    """RCCL communicator owned by liblas_hip.so (``las_comm_uid`` / ``las_comm_init`` / ``las_allreduce_f32``,
    include/las_hip.h): the gradient exchange a non-PyTorch host would bind.  The 128-byte unique id travels over
    whatever host channel the launcher has; ``from_process_group`` uses the existing torch.distributed group for that
    and nothing else."""

    def __init__(self, rank, world, uid: bytes, device=None):
        from . import _cabi
        self._cabi = _cabi
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            _cabi.check(_cabi.lib().las_comm_init(self.rank, self.world, uid))

    @staticmethod
    def new_uid() -> bytes:
        import ctypes
        from . import _cabi
        buf = ctypes.create_string_buffer(128)
        _cabi.check(_cabi.lib().las_comm_uid(buf))
        return buf.raw

    @classmethod
    def from_process_group(cls, device=None):
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.new_uid() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0], device)

    def allreduce_(self, flat, average=True):
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()):
            raise RuntimeError("las_allreduce_f32 needs a contiguous fp32 device buffer")
        with torch.cuda.device(flat.device):
            self._cabi.check(self._cabi.lib().las_allreduce_f32(flat.data_ptr(), flat.numel(), int(average),
                                                                self._cabi.stream_ptr()))
        return flat

    def destroy(self):
        with torch.cuda.device(self.device):
            self._cabi.check(self._cabi.lib().las_comm_destroy())


def sync_coin(seed_if_rank0=None):
    """Make the next draw of NumPy's GLOBAL stream — the per-forward teacher-forcing coin, one
    ``np.random.random_sample()`` in ``Speller.forward`` (las_model.py:189) — identical on every rank: rank 0 broadcasts
    its generator state and the other ranks install it.  Returns a token for ``restore_coin`` (the rank's own state, None
    on rank 0 / outside a multi-rank group), so that a rank's own use of ``np.random`` (shuffling, augmentation) keeps
    its private stream.  ``solver.batch_iterator`` calls the pair around the model call, and only when the coin matters
    (training with 0 < tf_rate < 1).  ``seed_if_rank0`` (tests): reseed rank 0's stream first."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    rank = dist.get_rank()
    if rank == 0 and seed_if_rank0 is not None:
        np.random.seed(int(seed_if_rank0))
    own = np.random.get_state()
    t = torch.zeros(625, dtype=torch.int64)
    if rank == 0:
        t[:624] = torch.from_numpy(own[1].astype(np.int64))
        t[624] = int(own[2])
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    if rank == 0:
        return None
    host = t.cpu().numpy()
    np.random.set_state((own[0], host[:624].astype(np.uint32), int(host[624]), 0, 0.0))
    return own


def restore_coin(token):
    """Give a rank its own NumPy stream back after the synchronised coin has been drawn (advanced by the one draw the
    forward consumed, as it would have been without the synchronisation)."""
    if token is not None:
        np.random.set_state(token)
        np.random.random_sample()


def shard_batch(n_items, rank, world):
    """Contiguous equal shards: rank r gets [r*n/world, (r+1)*n/world)."""
    assert n_items % world == 0, "global batch must divide evenly across ranks (equal shards keep the mean-loss gradient exact)"
    per = n_items // world
    return slice(rank * per, (rank + 1) * per)
