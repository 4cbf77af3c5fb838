"""One-process-per-GPU data parallelism over RCCL/xGMI (replaces the reference's single-process
``torch.nn.DataParallel``, pretrain_trainer.py:21 / clustering_trainer.py:25).

Encounters are sharded by rank; parameters and optimizer state are replicated.  Per joint step the
exchanges are: the small forward statistics that couple the batch (reconstruction
SSE + mask count, DEC column sums, KL sum, BatchNorm moments) and ONE flat gradient bucket.  All
are sum all-reduces; with ``backend='nccl'`` they run on RCCL, with ``'gloo'`` on CPU (tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as td


def is_sharded() -> bool:
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def world_size() -> int:
    return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1


def rank() -> int:
    return td.get_rank() if (td.is_available() and td.is_initialized()) else 0


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum all-reduce; a no-op when not sharded."""
    if is_sharded():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env vars (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    lr = int(os.environ.get('LOCAL_RANK', '0'))
    if ws > 1 and not td.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(lr)
            td.init_process_group(backend, rank=rk, world_size=ws, device_id=torch.device('cuda', lr))
        else:
            td.init_process_group(backend, rank=rk, world_size=ws)
    return rk, ws, lr


def shard_bounds(n, rk=None, ws=None):
    """Contiguous shard [lo, hi) of n items for this rank (rank r gets rows [r*n/P, (r+1)*n/P))."""
    rk = rank() if rk is None else rk
    ws = world_size() if ws is None else ws
    return (n * rk) // ws, (n * (rk + 1)) // ws


class FlatParams:
    """Re-homes a module's parameters (and their grads) into ONE flat f32 buffer each, so the
    per-step gradient exchange is a single contiguous all-reduce (2.33 MB at K=4) with no packing
    copies, and gradient clipping is one norm over one buffer."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('module has no trainable parameters')
        dev, total = self.params[0].device, sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self._attach(copy=True)

    def _attach(self, copy=False):
        o = 0
        for p in self.params:
            n = p.numel()
            if copy:
                self.flat[o:o + n].copy_(p.data.reshape(-1))
            if copy or p.data.data_ptr() != self.flat[o:o + n].data_ptr():
                p.data = self.flat[o:o + n].view_as(p.data)
            if p.grad is None or p.grad.data_ptr() != self.grad[o:o + n].data_ptr():
                p.grad = self.grad[o:o + n].view_as(p.data)
            o += n

    def zero_grad(self):
        self.grad.zero_()
        self._attach()          # re-attach in case something replaced .data / .grad

    def all_reduce_grads(self):
        """Sum over ranks: each rank's loss is already normalised by GLOBAL batch statistics, so
        the sum (not the mean) of rank gradients is the gradient of the global-batch loss."""
        all_reduce_sum_(self.grad)

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ semantics (pretrain_trainer.py:228) on the flat bucket."""
        total = torch.linalg.vector_norm(self.grad)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.grad.mul_(coef)
        return total

    def broadcast_(self, src: int = 0):
        if is_sharded():
            td.broadcast(self.flat, src=src)
