"""Data-parallel gradient exchange for the LAS hot path: one process per GPU, parameters replicated,
utterances of the minibatch sharded by rank, and ONE all-reduce of a flat fp32 gradient buffer per step
(RCCL over xGMI when the backend is "nccl").  Replaces the reference's ``nn.DataParallel`` wrapper
(train.py:76-78: per-step parameter broadcast + output gather + gradient reduce to GPU 0 in GIL-bound threads).

The batch-coupled quantities of the reference step are (solver/solver.py:43,96; las_model.py:189):
  * the batch-mean loss  -> average of equal-shard rank gradients == full-batch gradient,
  * the global grad-norm clip -> applied AFTER the all-reduce, identically on every rank,
  * the single teacher-forcing coin flip -> ``sync_coin`` broadcasts rank 0's NumPy RNG state draw.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


class FlatGradAllReducer:
    """Makes every ``p.grad`` a view into one contiguous fp32 buffer so the whole gradient crosses the fabric as a
    single collective (message: S 7.96 MB, P 39.9 MB)."""

    def __init__(self, module: torch.nn.Module, force: bool = False, direct: bool = False, comm=None):
        self.force = force          # all-reduce even in a 1-rank group (exercises the collective path)
        self.comm = comm            # optional CabiComm: the collective goes through liblas_hip's las_allreduce_f32
        self.direct = direct
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
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
        for p in self.params:
            if direct:
                p._las_direct_base = base
            elif hasattr(p, "_las_direct_base"):
                del p._las_direct_base
        module._las_flat_reducer = self      # solver.batch_iterator picks it up (zero / all-reduce / clip on the flat buffer)

    def zero(self):
        """Use instead of ``optimizer.zero_grad()`` (which would drop the views with set_to_none=True)."""
        self.flat.zero_()

    def check_views(self):
        """Every ``p.grad`` must still be its view of the flat buffer: ``optimizer.zero_grad()`` with torch's default
        ``set_to_none=True`` silently replaces them, after which the collective would reduce a buffer of zeros while
        the optimizer steps on unsynchronised local gradients.  Fail loudly instead."""
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                raise RuntimeError("FlatGradAllReducer: a parameter's .grad no longer aliases the flat gradient buffer "
                                   "(use reducer.zero() or optimizer.zero_grad(set_to_none=False), not zero_grad())")

    def allreduce_mean(self):
        """Average the flat gradient over the ranks with ONE collective (RCCL ``ncclAvg`` when the backend is nccl)."""
        self.check_views()
        if self.comm is not None:
            if self.comm.world > 1 or self.force:
                self.comm.allreduce_(self.flat, average=True)
            return
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size()
            if world > 1 or self.force:
                if dist.get_backend() == "nccl":
                    dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)
                else:                                   # gloo has no AVG
                    dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                    self.flat.div_(world)

    def clip_(self, max_norm=1.0):
        """clip_grad_norm_(params, max_norm) on the flat buffer (solver/solver.py:96), same formula as torch's."""
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
    """Keep the per-forward teacher-forcing coin (one ``np.random.random_sample()``, las_model.py:189) identical on
    every rank: rank 0 draws a 32-bit seed and everyone reseeds NumPy's global RNG with it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.zeros(1, dtype=torch.int64, device=dev)
    if dist.get_rank() == 0:
        t[0] = int(np.random.randint(0, 2 ** 31 - 1)) if seed_if_rank0 is None else int(seed_if_rank0)
    dist.broadcast(t, src=0)
    np.random.seed(int(t.item()))


def shard_batch(n_items, rank, world):
    """Contiguous equal shards: rank r gets [r*n/world, (r+1)*n/world)."""
    assert n_items % world == 0, "global batch must divide evenly across ranks (equal shards keep the mean-loss gradient exact)"
    per = n_items // world
    return slice(rank * per, (rank + 1) * per)
