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

    def __init__(self, module: torch.nn.Module, force: bool = False, direct: bool = False):
        self.force = force          # all-reduce even in a 1-rank group (exercises the collective path)
        # direct=True: the HIP backward kernels write each parameter gradient straight into its view of the flat buffer
        # (no per-parameter AccumulateGrad add kernel).  Requires one use of every parameter per backward and one
        # backward per step — what solver/solver.py:95-97 does.  See las_model.DIRECT_GRAD_WRITE.
        if direct:
            from .model import las_model
            las_model.DIRECT_GRAD_WRITE = True
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.numel = n

    def zero(self):
        """Use instead of ``optimizer.zero_grad()`` (which would drop the views with set_to_none=True)."""
        self.flat.zero_()

    def check_views(self):
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            assert p.grad is not None and p.grad.untyped_storage().data_ptr() == base, "a .grad left the flat buffer"

    def allreduce_mean(self):
        if dist.is_available() and dist.is_initialized():
            if dist.get_world_size() > 1 or self.force:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                self.flat.div_(dist.get_world_size())

    def clip_(self, max_norm=1.0):
        """clip_grad_norm_(params, max_norm) on the flat buffer (solver/solver.py:96), same formula as torch's."""
        total = torch.linalg.vector_norm(self.flat)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.flat.mul_(coef)
        return total


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
