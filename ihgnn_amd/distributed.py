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


def force_collectives() -> bool:
    """``IHG_FORCE_COLLECTIVES=1``: run every collective even in a ONE-rank process group (they are identities there).  This is how the RCCL
    code paths - ``ReduceOp.AVG``, ``reduce_scatter_tensor``, ``all_gather_into_tensor``, async launches from gradient hooks - are
    executed on a box with a single GPU (tests/test_gpu_parity.py, tools/two_rank_check.py --ranks 1 --backend nccl)."""
    return os.environ.get('IHG_FORCE_COLLECTIVES') == '1'


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from torchrun's environment.  -> (rank, local_rank, world_size)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            # IHG_DIST_BACKEND=gloo: the driver's ranks on ONE GPU (RCCL refuses two ranks per device): how the multi-rank training loop is tested on a single-GPU box
            backend = os.environ.get('IHG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl' and local < torch.cuda.device_count():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradientSync:
    """Flat gradient buffer + one averaging all-reduce per step."""

    owns_optimizer = False

    def __init__(self, parameters: Iterable[torch.nn.Parameter], group=None):
        everything = list(parameters)
        self.params: List[torch.nn.Parameter] = [p for p in everything if p.requires_grad]
        # position of every exchanged parameter in the caller's full list (frozen ones included): torch.optim.Adam(model.parameters()) numbers its state entries that way
        self.param_index: List[int] = [i for i, p in enumerate(everything) if p.requires_grad]
        self.n_all_params = len(everything)
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.distributed = dist.is_initialized() and (self.world_size > 1 or force_collectives())      # collectives are issued
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
        if self.distributed:
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
        if self.distributed:
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.group)


class BucketedGradientSync(GradientSync):
    """``GradientSync`` whose flat buffer is cut into buckets, each all-reduced from RCCL's own stream as soon as the backward has
    produced it, so the exchange overlaps the rest of the backward instead of starting after its last kernel.

    Bucket order = the order the backward finishes them: first the dense parameters (prediction bias, layer weights - ready while
    the propagation's backward is still running), then one bucket per embedding table (produced by the very last kernels of the
    step; 99 % of the bytes, so they stay exposed - see DESIGN.md §7 for the expected exposed time per config).  A bucket is
    launched by the ``post_accumulate_grad`` hook of its last parameter (``async_op=True``: the collective waits for the
    producing kernels through an event, not the host); ``average_gradients`` waits for all of them.  Result identical to the flat
    all-reduce (same sums, element by element)."""

    def __init__(self, named_parameters, group=None):
        named = [(n, p) for n, p in named_parameters if p.requires_grad]
        tables = [(n, p) for n, p in named if '.embedding' in n or n.startswith('embedding')]
        dense = [(n, p) for n, p in named if (n, p) not in tables]
        ordered = dense + tables
        super().__init__([p for _, p in ordered], group)
        self.buckets = []                                    # (begin, end) element ranges of the flat buffer
        offset = 0
        groups = ([dense] if dense else []) + [[t] for t in tables]
        self._bucket_of = {}
        for index, members in enumerate(groups):
            n = sum(p.numel() for _, p in members)
            self.buckets.append((offset, offset + n))
            offset += n
            for _, p in members:
                self._bucket_of[p] = index
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._launched = [False] * len(self.buckets)
        self._arm()
        if self.distributed:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _arm(self) -> None:
        counts = [0] * len(self.buckets)
        for p in self.params:
            counts[self._bucket_of[p]] += 1
        self._pending = counts
        self._launched = [False] * len(self.buckets)
        self._works = []

    def _launch(self, index: int) -> None:
        begin, end = self.buckets[index]
        piece = self.flat[begin:end]
        if self._use_avg is None:
            self._use_avg = dist.get_backend(self.group) == 'nccl'
        work = None
        if self._use_avg:
            try:
                work = dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            except (RuntimeError, ValueError):               # a build without ncclAvg says so before anything is enqueued
                self._use_avg = False
        if work is None:
            work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, piece, self._use_avg))
        self._launched[index] = True

    def _on_grad(self, p) -> None:
        index = self._bucket_of[p]
        if self._launched[index]:
            # a second backward before average_gradients() (gradient accumulation, two losses): the bucket's exchange is already in flight,
            # the new contribution would land on top of the averaged values and never be reduced - the ranks would drift apart silently
            raise RuntimeError('BucketedGradientSync: a gradient arrived for a bucket whose all-reduce was already launched; call '
                               'average_gradients() after every backward, or use the flat exchange for gradient accumulation')
        view_ok = p.grad is not None and self.flat.data_ptr() <= p.grad.data_ptr() < self.flat.data_ptr() + self.flat.numel() * self.flat.element_size()
        if not view_ok:                                      # someone detached the view (set_to_none): exchange everything at the end
            self._pending[index] = -1
            return
        self._pending[index] -= 1
        if self._pending[index] == 0:
            self._launch(index)

    def average_gradients(self):
        if self.flat is None or not self.distributed:
            return
        self._reattach()
        for index in range(len(self.buckets)):               # buckets whose hooks did not all fire (unused parameters, detached views)
            if not self._launched[index]:
                self._launch(index)
        for work, piece, averaged in self._works:
            work.wait()                                      # the current stream waits for the collective; the host does not
            if not averaged:
                piece.div_(self.world_size)
        self._arm()


class _ShardOptimizer:
    def __init__(self, sync: 'ShardedGradientSync', lr: float, weight_decay: float):
        self.sync = sync
        shard = torch.nn.Parameter(sync.param_shard)
        shard.grad = sync.grad_shard
        self.shard = shard
        if shard.is_cuda:
            from .optim import Adam
            self.inner = Adam([shard], lr, weight_decay=weight_decay)
        else:
            self.inner = torch.optim.Adam([shard], lr, weight_decay=weight_decay)
        self.param_groups = self.inner.param_groups

    def step(self):
        self.shard.grad = self.sync.grad_shard
        self.inner.step()
        self.sync.gather_parameters()

    def zero_grad(self, set_to_none: bool = False):
        self.sync.zero_grad()

    def _gather_full(self):
        """``{exp_avg, exp_avg_sq: full-length flat vectors, step}`` gathered from every rank's shard (COLLECTIVE), or ``None`` before the first step."""
        sync = self.sync
        entry = next(iter(self.inner.state_dict()['state'].values()), None)
        if entry is None:
            return None
        full = {}
        for key in ('exp_avg', 'exp_avg_sq'):
            gathered = torch.empty(sync.padded, dtype=sync.param_shard.dtype, device=sync.param_shard.device)
            if sync.distributed:
                dist.all_gather_into_tensor(gathered, entry[key].reshape(-1).contiguous(), group=sync.group)
            else:
                gathered.copy_(entry[key].reshape(-1))
            full[key] = gathered[:sync.flat.numel()].clone()
        full['step'] = entry['step']
        return full

    def state_dict(self):
        """Adam state of ALL parameters in ``torch.optim.Adam``'s own layout - ``{'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]}``, parameter i = the i-th of
        ``model.parameters()``, frozen parameters counted - what the reference writes (``Main.py:255-259``) and reads (``Main.py:208-212``) and what the flat / bucketed modes' optimizers write and read: a checkpoint taken under one
        ``--grad_sync`` mode (or world size, or by the reference) resumes under another.  Every rank's shard of ``exp_avg`` / ``exp_avg_sq`` is gathered, so rank 0's file is complete.

        COLLECTIVE: it issues ``all_gather_into_tensor`` - EVERY rank of the group must call it, at the same point of the program.  The
        usual ``if chief: torch.save({... optimizer.state_dict()})`` would leave rank 0 waiting for peers that never come; call
        :func:`checkpoint_state` on every rank and save its result on the chief (``Main.py`` does).  ``local_state_dict()`` is the
        collective-free form (this rank's shard only)."""
        sync = self.sync
        full = self._gather_full()
        groups = [dict(g) for g in self.inner.state_dict()['param_groups']]
        for g in groups:
            g['params'] = list(range(sync.n_all_params))
        state = {}
        if full is not None:
            offset = 0
            for index, p in zip(sync.param_index, sync.params):
                n = p.numel()
                state[index] = {'step': full['step'].clone() if torch.is_tensor(full['step']) else full['step'],
                                'exp_avg': full['exp_avg'][offset:offset + n].view_as(p).clone(), 'exp_avg_sq': full['exp_avg_sq'][offset:offset + n].view_as(p).clone()}
                offset += n
        return {'state': state, 'param_groups': groups}

    def local_state_dict(self):
        """This rank's shard of the Adam state, no collective (a per-rank checkpoint file; not loadable at another world size)."""
        return {'adam_shard': self.inner.state_dict(), 'shard_range': tuple(self.sync.shard_range), 'numel': self.sync.flat.numel()}

    def load_state_dict(self, state):
        """Takes ``torch.optim.Adam``'s layout (``state_dict()`` above; a checkpoint of the flat / bucketed modes or of the reference) and the flat form rounds 3 - 4 wrote
        (``{'sharded_adam': ..., 'numel': ...}``)."""
        sync = self.sync
        if 'sharded_adam' in state:
            if state['numel'] != sync.flat.numel():
                raise ValueError(f"checkpoint holds Adam state for {state['numel']} parameters, the model has {sync.flat.numel()}")
            full = state['sharded_adam']
        elif 'state' in state and 'param_groups' in state:
            entries = state['state']
            if not entries:
                full = {}
            else:
                # entries are keyed by the parameter's position in model.parameters() (torch.optim.Adam's numbering).  A parameter that never received a gradient has
                # no entry there (Adam creates state at a parameter's first step): it gets zero moments and the common step count; a frozen one is not exchanged at all
                stray = sorted(k for k in entries if k not in set(sync.param_index))
                if stray:
                    raise ValueError(f'checkpoint holds Adam state for parameter(s) {stray} that this model does not train (frozen or out of range: {sync.n_all_params} parameters)')
                steps = {float(e['step']) for e in entries.values()}
                if len(steps) != 1:
                    raise ValueError(f'the sharded optimizer keeps ONE step count; the checkpoint holds {sorted(steps)}')
                ordered = []
                for index, p in zip(sync.param_index, sync.params):
                    e = entries.get(index)
                    if e is None:
                        ordered.append({'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)})
                    elif tuple(e['exp_avg'].shape) != tuple(p.shape):
                        raise ValueError(f"checkpoint Adam state {index} of shape {tuple(e['exp_avg'].shape)} does not fit parameter of shape {tuple(p.shape)}")
                    else:
                        ordered.append(e)
                device = sync.param_shard.device
                full = {'exp_avg': torch.cat([e['exp_avg'].reshape(-1).to(device) for e in ordered]), 'exp_avg_sq': torch.cat([e['exp_avg_sq'].reshape(-1).to(device) for e in ordered]),
                        'step': next(iter(entries.values()))['step']}
        else:
            raise ValueError('not an Adam checkpoint (expected torch.optim.Adam\'s state_dict layout or the dict written by _ShardOptimizer.state_dict of rounds 3 - 4)')
        if not full:
            return
        lo, hi = sync.shard_range
        inner = self.inner.state_dict()
        if not inner['state']:                               # no step taken yet: let the inner optimizer create its buffers
            self.shard.grad = torch.zeros_like(self.shard)
            saved = self.shard.detach().clone()
            self.inner.step()
            with torch.no_grad():
                self.shard.copy_(saved)
            inner = self.inner.state_dict()
        entry = next(iter(inner['state'].values()))
        for key in ('exp_avg', 'exp_avg_sq'):
            padded = torch.zeros(sync.padded, dtype=sync.param_shard.dtype, device=sync.param_shard.device)
            padded[:sync.flat.numel()].copy_(full[key].to(padded.device))
            entry[key] = padded[lo:hi].clone().view_as(entry[key])
        entry['step'] = full['step'].clone() if torch.is_tensor(full['step']) else torch.tensor(float(full['step']))
        for mine, theirs in zip(inner['param_groups'], state['param_groups']):
            mine.update({k: v for k, v in theirs.items() if k != 'params'})
        self.inner.load_state_dict(inner)


class ShardedGradientSync(GradientSync):
    """Reduce-scatter -> Adam on this rank's 1/W shard -> all-gather of the parameters (for config C5's ~10 GB of
    gradients): the same bytes cross xGMI as in an all-reduce, but each rank keeps and updates only 1/W of the Adam state
    (2 x 10 GB -> 2 x 1.25 GB at W = 8) and streams 1/W of the update.  Parameters live in one flat buffer (module parameters are
    views of it), padded to a multiple of W.  Adam is elementwise, so the result equals the replicated update bit for bit given
    the same averaged gradient.  ``optimizer(lr)`` returns the step object to use in place of ``Adam(model.parameters())``."""

    owns_optimizer = True

    def __init__(self, parameters, group=None):
        super().__init__(parameters, group)
        w = self.world_size
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        total = self.flat.numel()
        self.padded = -(-total // w) * w
        ref = self.params[0]
        grads = torch.zeros(self.padded, dtype=ref.dtype, device=ref.device)
        self.flat = grads[:total]
        self._grads_padded = grads
        self.flat_params = torch.zeros(self.padded, dtype=ref.dtype, device=ref.device)
        offset = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat_params[offset:offset + n].copy_(p.data.reshape(-1))
                p.data = self.flat_params[offset:offset + n].view_as(p)
                p.grad = self.flat[offset:offset + n].view_as(p)
                offset += n
        share = self.padded // w
        self.shard_range = (rank * share, (rank + 1) * share)
        self.param_shard = self.flat_params[self.shard_range[0]:self.shard_range[1]]
        self.grad_shard = torch.zeros(share, dtype=ref.dtype, device=ref.device)

    def optimizer(self, lr: float, weight_decay: float = 0.0) -> _ShardOptimizer:
        return _ShardOptimizer(self, lr, weight_decay)

    def average_gradients(self):
        self._reattach()
        lo, hi = self.shard_range
        if not self.distributed:
            self.grad_shard.copy_(self._grads_padded[lo:hi])
            return
        backend = dist.get_backend(self.group)
        if backend == 'nccl':
            try:
                dist.reduce_scatter_tensor(self.grad_shard, self._grads_padded, op=dist.ReduceOp.AVG, group=self.group)
                return
            except (RuntimeError, ValueError):
                pass
        if backend == 'nccl':
            dist.reduce_scatter_tensor(self.grad_shard, self._grads_padded, op=dist.ReduceOp.SUM, group=self.group)
            self.grad_shard.div_(self.world_size)
            return
        dist.all_reduce(self._grads_padded, op=dist.ReduceOp.SUM, group=self.group)      # gloo (CPU tests): no reduce-scatter
        torch.div(self._grads_padded[lo:hi], self.world_size, out=self.grad_shard)

    def gather_parameters(self) -> None:
        if self.distributed:
            dist.all_gather_into_tensor(self.flat_params, self.param_shard.clone(), group=self.group)

    def zero_grad(self) -> None:
        self._grads_padded.zero_()

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.distributed:
            dist.broadcast(self.flat_params, src=src, group=self.group)


class CotangentSync:
    """Data parallelism WITHOUT a dense gradient exchange: the ranks exchange the cotangents of the batch rows.

    Every rank's full-graph propagation ``F(theta) [N, D]`` is the same function of the same parameters (that is what the replicas are), the batch tail of rank r reads
    ``F`` at its 3B batch rows only (``Models/RawGnn.py:128-131`` of the reference), and the propagation's backward is LINEAR in the cotangent ``dL/dF``.  The averaged
    gradient of all parameters is therefore ``J^T (1/W sum_r dL_r/dF)`` plus the sparse ``items_bias`` gradient: each rank computes its ``[3B, D + 1]`` row gradients
    (scaled by 1 / W), ONE all-gather brings everybody's (``3B (D + 1) 4`` bytes per rank: 6.8 MB at config C3, 10 MB at C5 - the dense gradients are 194.6 MB and
    9.42 GB), duplicates are summed in a fixed order by the combine kernel, and every rank runs the same propagation backward on the union.  No all-reduce; Adam is
    the plain local one; the replicas stay identical because the kernels are bitwise deterministic and every rank sees the gathered rows in rank order
    (``check_replicas()`` verifies it when asked to).

    Cost against the dense exchange: the last layer's masked pull and the taps see ``3 B W`` rows instead of ``3 B``.  Use: ``model.bce_loss(u, q, i, y,
    cotangent_sync=sync)``, ``loss.backward()``, ``optimizer.step()``, ``sync.zero_grad()``; ``average_gradients()`` has nothing left to do.  Every rank calls
    ``bce_loss`` once per step (its two all-gathers are collectives); batches of different length are padded inside (``gather_rows``)."""

    owns_optimizer = False
    mode = 'cotangent'

    def __init__(self, parameters: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in parameters if p.requires_grad]
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.distributed = dist.is_initialized() and (self.world_size > 1 or force_collectives())
        self.flat = None
        self.exchanged_bytes = 0                              # bytes this rank RECEIVED in the last step's exchanges (rows + row gradients of the other ranks)
        self.sent_bytes = 0
        self.equal_batches = False                            # True: the caller promises every rank's batch has the same size (no size exchange per step)
        self.time_exchanges = False                           # True: HIP events around every all-gather on the step's stream land in exchange_events (bench.py)
        self.exchange_events = []

    # -- the two collectives of a step ----------------------------------------------------------
    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """``[W, *t.shape]``: every rank's ``t`` in rank order."""
        t = t.contiguous()
        if not self.distributed:
            return t.unsqueeze(0)
        out = torch.empty((self.world_size,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        timed = self.time_exchanges and t.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_gather_into_tensor(out.view(-1), t.view(-1), group=self.group)      # (flat on both sides: gloo accepts nothing else)
        if timed:
            e1.record()
            self.exchange_events.append((e0, e1))
        self.exchanged_bytes += (self.world_size - 1) * t.numel() * t.element_size()
        self.sent_bytes += t.numel() * t.element_size()
        return out

    def gather_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """The union of the ranks' batch node rows, ``[3 B W]`` int64 laid out as ``users of rank 0 .. W-1 | queries ... | items ...`` (the thirds address disjoint node
        ranges: what the combine kernel's duplicate search relies on), with ``.as_int32`` beside it.  Called before the forward: the last layer computes / is
        differentiated at these rows.  Ranks whose batches are shorter than the longest (the sharded sampler's differ by a row; ``equal_batches = True`` promises they
        never do and saves the size exchange, a host round trip) repeat their last row here and contribute exact zeros for it in ``exchange``."""
        self.exchanged_bytes = self.sent_bytes = 0
        n = int(rows.shape[0])
        if n % 3 or n == 0:
            raise ValueError('batch node rows come as users | queries | items, at least one of each')
        b = n // 3
        longest = b
        if self.distributed and not self.equal_batches:
            t = torch.tensor([b], dtype=torch.int64, device=rows.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            longest = int(t.item())
        thirds = rows.view(3, b)
        if longest > b:
            thirds = torch.cat([thirds, thirds[:, -1:].expand(3, longest - b)], 1)
        self._local_rows, self._padded_rows = b, longest
        everyone = self._all_gather(thirds)                                  # [W, 3, B]
        union = everyone.permute(1, 0, 2).reshape(-1).contiguous()
        union.as_int32 = union.to(torch.int32)
        self._union_rows = union
        return union

    def exchange(self, rows: torch.Tensor, rowgrad: torch.Tensor):
        """``(rows, rowgrad)`` of this rank's batch tail -> the union's, in ``gather_rows``' order (called by the batch tail's backward, ``ops._HemBceLoss``)."""
        n, width = int(rowgrad.shape[0]), int(rowgrad.shape[1])
        union_rows = getattr(self, '_union_rows', None)
        if union_rows is None or n != 3 * self._local_rows:
            raise RuntimeError('CotangentSync.exchange: gather_rows() was not called for this batch (use model.bce_loss(..., cotangent_sync=sync))')
        mine = rowgrad.view(3, self._local_rows, width)
        if self._padded_rows > self._local_rows:
            mine = torch.cat([mine, mine.new_zeros(3, self._padded_rows - self._local_rows, width)], 1)
        everyone = self._all_gather(mine)                                    # [W, 3, B, C]
        union = everyone.permute(1, 0, 2, 3).reshape(-1, width).contiguous()
        self._union_rows = None
        return union_rows, union

    # -- the GradientSync surface the training loops use ------------------------------------------
    def average_gradients(self) -> None:
        """Nothing to do: the backward already produced the gradient of the mean of the ranks' losses, identically on every rank."""

    def zero_grad(self) -> None:
        for p in self.params:
            p.grad = None

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.distributed:
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.group)

    def check_replicas(self) -> float:
        """Largest absolute difference of any parameter element between this rank and rank 0 (a collective; 0.0 is the contract: the replicas take bitwise
        identical steps).  For tests and an occasional assertion in long runs - not part of a step."""
        worst = 0.0
        if not self.distributed:
            return worst
        for p in self.params:
            ref = p.data.clone()
            dist.broadcast(ref, src=0, group=self.group)
            worst = max(worst, float((ref - p.data).abs().max()))
        t = torch.tensor([worst], dtype=torch.float64, device=self.params[0].device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())


def checkpoint_state(epoch: int, model: torch.nn.Module, optimizer) -> dict:
    """The dict a training checkpoint holds (written at ``Main.py:255-259`` of the reference, read at ``Main.py:208-212``: epoch, model, optimizer).  Call it on EVERY rank: a sharded
    optimizer's ``state_dict()`` gathers the ranks' Adam shards with a collective, and a chief-only call would deadlock.  Every rank gets the
    same complete dict back; only the chief writes it."""
    return {'epoch_count': int(epoch), 'model': model.state_dict(), 'optimizer': optimizer.state_dict()}


def make_gradient_sync(model: torch.nn.Module, mode: str = 'bucketed', group=None) -> GradientSync:
    """``flat`` | ``bucketed`` | ``sharded`` dense-gradient exchange, or ``cotangent`` (batch-row cotangents instead of gradients), for a model (see the classes above)."""
    if mode == 'flat':
        return GradientSync(model.parameters(), group)
    if mode == 'bucketed':
        return BucketedGradientSync(model.named_parameters(), group)
    if mode == 'sharded':
        return ShardedGradientSync(model.parameters(), group)
    if mode == 'cotangent':
        return CotangentSync(model.parameters(), group)
    raise ValueError(f'unknown gradient sync mode {mode!r}')


def cotangent_bytes_per_rank(batch_rows: int, feature_width: int) -> int:
    """Bytes one rank hands to the cotangent exchange per step: the 3 B int64 node rows of its batch and their ``[D + 4]`` float row gradients (``D`` = the width of the
    propagation's output, d (L + 1); column D carries the items-bias gradient)."""
    return 3 * int(batch_rows) * (8 + 4 * (int(feature_width) + 4))


def choose_gradient_sync(gradient_bytes: int, world_size: int, fused_loss: bool = True, cotangent_bytes: Optional[int] = None) -> str:
    """The exchange a run takes when it is not told one, by the bytes a rank receives per step: ``cotangent`` where the model trains through the fused batch tail (IHGNN /
    HGCN layers + HEM + BCE: every configuration of BASELINE.json) and the other ranks' row cotangents - ``(W - 1) x cotangent_bytes`` - are fewer than the
    ``2 (W - 1) / W x gradient_bytes`` of a ring all-reduce (C2 - C5: 7 - 230 times fewer; a toy model like C1, whose whole gradient is 0.7 MB, keeps the dense exchange);
    otherwise ``bucketed`` while the flat gradient fits comfortably beside the model (it overlaps the dense bucket with the backward) and ``sharded`` beyond 2 GiB (config C5:
    9.4 GB of gradients, 2 x 9.4 GB of Adam state per replica -> 1 / W of it)."""
    if fused_loss and (cotangent_bytes is None or cotangent_bytes * max(world_size, 1) < 2 * gradient_bytes):
        return 'cotangent'
    return 'sharded' if gradient_bytes > (2 << 30) else 'bucketed'


class ShardedBatchSampler:
    """Batch sampler of the data-parallel training loop: every epoch is ONE permutation of the dataset shared by all ranks
    (a function of ``seed`` and the epoch), rank r takes positions ``r, r + W, r + 2W, ...`` of it, and every rank cuts its
    share into the SAME number of batches (sizes differ by at most one row), so the ranks issue the same number of gradient
    all-reduces and the union of all ranks' batches covers the epoch exactly once - no padding, no duplicates.

    With one rank and ``shuffle=True`` this is ``DataLoader(batch_size=B, shuffle=True)``'s batching (full batches, a shorter
    last one) over its own permutation.  Global batch = ``B * W`` rows per step (the learning rate is left as configured)."""

    def __init__(self, n_items: int, batch_size: int, rank: int = 0, world_size: int = 1, shuffle: bool = True, seed: int = 0):
        if n_items < world_size:
            raise ValueError(f'{n_items} training rows cannot be sharded over {world_size} ranks')
        self.n_items, self.batch_size, self.rank, self.world_size = int(n_items), int(batch_size), int(rank), int(world_size)
        self.shuffle, self.seed, self.epoch = bool(shuffle), int(seed), 0
        largest_share = -(-self.n_items // self.world_size)
        self.n_batches = -(-largest_share // self.batch_size)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return self.n_batches

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed * 1_000_003 + self.epoch)
            order = torch.randperm(self.n_items, generator=g)
        else:
            order = torch.arange(self.n_items)
        mine = order[self.rank::self.world_size].tolist()
        if self.world_size == 1:
            for lo in range(0, len(mine), self.batch_size):
                yield mine[lo:lo + self.batch_size]
            return
        if len(mine) < self.n_batches:
            raise ValueError(f'rank {self.rank} holds {len(mine)} rows for {self.n_batches} batches: batch_size {self.batch_size} is too small for '
                             f'{self.world_size} ranks (an empty batch on one rank would leave the others waiting in the gradient exchange)')
        base, extra = divmod(len(mine), self.n_batches)
        lo = 0
        for b in range(self.n_batches):
            hi = lo + base + (1 if b < extra else 0)
            yield mine[lo:hi]
            lo = hi


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
