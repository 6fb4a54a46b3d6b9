"""Data-parallel training across the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-process.  The scaling contract of this build:
every rank holds a full replica of the parameters and of the hypergraph layout, runs the full-graph
propagation on its own mini-batch, and ONE all-reduce averages all gradients before Adam.  Every parameter is
touched densely each step (all embedding rows feed the propagation), so the gradients are packed into a single
flat fp32 buffer whose slices ARE the ``.grad`` tensors: the all-reduce runs in place on one large message
(xGMI rings are per-link bound - one big collective beats many small ones) with no pack / unpack copies.

``torch.distributed`` backend ``nccl`` is RCCL on ROCm; ``gloo`` is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from torchrun's environment.  -> (rank, local_rank, world_size)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl' and local < torch.cuda.device_count():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradientSync:
    """Flat gradient buffer + one averaging all-reduce per step."""

    def __init__(self, parameters: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in parameters if p.requires_grad]
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self._use_avg = None
        if not self.params:
            self.flat = None
            return
        ref = self.params[0]
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        offset = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[offset:offset + n].view_as(p)       # autograd accumulates in place into the slice
            offset += n

    def zero_grad(self) -> None:
        """Use instead of ``optimizer.zero_grad()`` (whose default set_to_none would detach the views)."""
        if self.flat is not None:
            self.flat.zero_()

    def _reattach(self) -> None:
        offset = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[offset:offset + n].view_as(p)
            if p.grad is None:
                p.grad = view
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            offset += n

    def average_gradients(self):
        """All-reduce (sum) the flat buffer and divide by the world size.  No-op on a single rank."""
        if self.flat is None:
            return
        self._reattach()
        if self.world_size > 1:
            if self._use_avg is None:
                self._use_avg = dist.get_backend(self.group) == 'nccl'
            if self._use_avg:                                # RCCL averages inside the collective: no second pass over the buffer
                try:
                    dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
                    return
                except (RuntimeError, ValueError):           # a build without ncclAvg says so before anything is enqueued
                    self._use_avg = False
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world_size)

    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every replica start from rank ``src``'s weights."""
        if self.world_size > 1:
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.group)


def shard_range(n_items: int, rank: int, world_size: int) -> range:
    """Contiguous, balanced slice of ``range(n_items)`` owned by ``rank`` (evaluation logs, batch rows)."""
    base, extra = divmod(n_items, world_size)
    begin = rank * base + min(rank, extra)
    return range(begin, begin + base + (1 if rank < extra else 0))


def all_reduce_sums(values: List[float], device: torch.device, group=None) -> List[float]:
    """Sum a short list of floats over ranks (metric totals + counts at evaluation)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(values)
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()
