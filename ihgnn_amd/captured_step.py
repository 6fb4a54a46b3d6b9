"""One training step of ``RawGnn`` recorded as a hipGraph and replayed (``torch.cuda.graph``).

A step of the hot path is ~85 kernel launches issued from Python through ctypes.  At the BASELINE workloads from C2 up the GPU is
the bound (the kernels of a C3 step add up to the step's wall time: profiles/r3), so recording buys nothing there; on small graphs
(config C1: 20,000 hyperedges, every kernel a few microseconds) the step is bound by the host's launch rate, and a replay removes
that.  What is recorded: forward (``RawGnn.bce_loss``), backward, and the Adam update with its two step-dependent scalars read from
device memory (``ihg_adam_step_device_scalars``).  Per replay the host copies the batch into the static input buffers, refreshes the two
Adam scalars and launches the graph.  Single process only (a recorded RCCL exchange is not attempted).
"""
from __future__ import annotations

import torch

from .optim import Adam


class CapturedTrainingStep:
    """``step(users, queries, items, labels) -> loss`` (a device scalar, valid until the next call), same result as::

        loss = model.bce_loss(users, queries, items, labels); loss.backward(); optimizer.step(); optimizer.zero_grad()

    for batches of ``batch_rows`` rows.  The learning rate may change between calls (it enters through the device scalars)."""

    def __init__(self, model, optimizer: Adam, batch_rows: int, warmup_batch=None):
        if not isinstance(optimizer, Adam):
            raise TypeError('CapturedTrainingStep records ihgnn_amd.optim.Adam')
        self.model, self.optimizer, self.batch_rows = model, optimizer, int(batch_rows)
        dev = next(model.parameters()).device
        self.users = torch.zeros(batch_rows, dtype=torch.int64, device=dev)
        self.queries = torch.zeros(batch_rows, dtype=torch.int64, device=dev)
        self.items = torch.zeros(batch_rows, dtype=torch.int64, device=dev)
        self.labels = torch.zeros(batch_rows, dtype=torch.float32, device=dev)
        self.scalars = torch.zeros(2, dtype=torch.float32, device=dev)
        self._unit = torch.ones((), dtype=torch.float32, device=dev)         # root gradient of every replay (autograd would fill a fresh one inside the recording)
        self._table, self._table_first, self._table_lr = None, 0, None   # Adam scalars of the coming steps, on the device
        if warmup_batch is not None:
            for dst, src in zip((self.users, self.queries, self.items, self.labels), warmup_batch):
                dst.copy_(src)
        optimizer.ensure_state()                             # exp_avg / exp_avg_sq exist before the recording (not in the graph's pool)
        params = [p for group in optimizer.param_groups for p in group['params'] if p.requires_grad]
        slots = []                                           # (module, attribute name, parameter) of every trainable parameter of the model
        wanted = {id(p) for p in params}
        for module in model.modules():
            for name, p in list(module._parameters.items()):
                if p is not None and id(p) in wanted:
                    slots.append((module, name, p))
        if {id(p) for _, _, p in slots} != wanted:
            raise ValueError('CapturedTrainingStep: the optimizer holds parameters that are not parameters of the model')
        first_slot = {}
        for module, name, p in slots:
            first_slot.setdefault(id(p), (module, name, p))
        slots_by_param = [first_slot[id(p)] for p in params]

        def forward_backward():
            # The forward runs on fresh LEAF ALIASES of the parameters (same storage, new autograd identity) and the gradients come from
            # torch.autograd.grad.  A parameter's AccumulateGrad node belongs to the stream that first used it and stays alive as long as the
            # caller holds ANY earlier loss; autograd then routes the parameter's gradient through that (default) stream - inside a
            # recording on another stream that is a cross-stream dependency the capture cannot hold (it ends in a crash at capture end).
            # Aliases made here have no history: their gradient edges are created on the recording's stream.
            aliases = {id(p): p.detach().requires_grad_(True) for p in params}
            try:
                for module, name, p in slots:
                    module._parameters[name] = aliases[id(p)]
                loss = model.bce_loss(self.users, self.queries, self.items, self.labels)
            finally:
                for module, name, p in slots:
                    module._parameters[name] = p
            grads = torch.autograd.grad(loss, [aliases[id(p)] for p in params], grad_outputs=self._unit, allow_unused=True)
            return loss, grads

        # torch's recipe for whole-step capture: a few eager iterations on a side stream (allocator warm-up), then the recording; the
        # gradients the recording allocates live in the graph's pool and are rewritten by every replay.  The warm-up runs forward and
        # backward only: no parameter changes.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                _, warm_grads = forward_backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        unused = [name for (module, name, p), g in zip(slots_by_param, warm_grads) if g is None]
        del warm_grads
        if unused:
            # the eager Adam.step skips a parameter whose gradient is None (no decay, no step count); a recording cannot - it would have to
            # step it with a zero gradient, which decays it under weight_decay > 0 and lets the step counts of the two paths drift apart
            raise ValueError(f'CapturedTrainingStep: parameters without a gradient in this step ({", ".join(unused)}): freeze them (requires_grad = False) '
                             'or train eagerly')
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, grads = forward_backward()
            for p, g in zip(params, grads):
                p.grad = g
            optimizer.launch_with_device_scalars(self.scalars)
        self._baked = self._baked_settings()

    def _baked_hyperparameters(self):
        """The optimizer's hyper-parameters and the model attribute a replay cannot change any more (cheap: compared at every ``step()``)."""
        group = self.optimizer.param_groups[0]
        return (tuple(group['betas']), float(group['eps']), float(group['weight_decay']), bool(self.model.batch_rows_only_last_layer))

    def _baked_settings(self):
        """Everything a replay cannot change any more: the recording holds the kernels these settings selected.  Only the learning rate is
        refreshed per replay (it enters through the device scalars)."""
        import os
        switches = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith('IHG_')))
        from . import ops
        # every path switch of ihgnn_amd.ops: its module-level UPPER_CASE scalars, found by name - a switch added to ops is baked in here without this list being touched
        # (COTANGENT_PLANES, TWO_HOP_MERGED, ... : tests and bench.py toggle these attributes between steps)
        flags = tuple((name, value) for name, value in sorted(vars(ops).items()) if name.isupper() and not name.startswith('_') and (value is None or isinstance(value, (bool, int, float, str))))
        return self._baked_hyperparameters() + (switches, flags)

    def stale(self, full: bool = False) -> bool:
        """True when a setting that is baked into the recording has changed since: the caller must record a new step (``TrainTestHelper`` does) - replaying would
        silently ignore the change.  The default compares Adam's betas / eps / weight decay and ``batch_rows_only_last_layer`` - four values, what every ``step()``
        checks; ``full=True`` also the ``IHG_*`` environment and the path switches of ``ihgnn_amd.ops`` (a sort over the environment: for the caller to ask once per
        epoch, not per replay on the launch-bound path the recording exists for)."""
        if full:
            return self._baked != self._baked_settings()
        return self._baked[:4] != self._baked_hyperparameters()

    TABLE_STEPS = 2048

    def _refresh_scalars(self) -> None:
        """``scalars <- (lr / (1 - beta1^t), sqrt(1 - beta2^t))`` of the coming step by a device-to-device copy out of a table that holds the
        next ``TABLE_STEPS`` steps (rebuilt when it runs out or the learning rate changes): no host buffer is read by an in-flight copy."""
        opt = self.optimizer
        group = opt.param_groups[0]
        t = opt.next_step()
        if self._table is None or self._table_lr != group['lr'] or not (self._table_first <= t < self._table_first + self.TABLE_STEPS):
            import numpy as np
            # exactly ihg_adam_step's arithmetic: lr and the betas arrive there as fp32, the bias corrections are formed in double
            lr, beta1, beta2 = (float(np.float32(x)) for x in (group['lr'], group['betas'][0], group['betas'][1]))
            steps = np.arange(t, t + self.TABLE_STEPS, dtype=np.float64)
            table = torch.from_numpy(np.stack([lr / (1.0 - np.power(beta1, steps)), np.sqrt(1.0 - np.power(beta2, steps))], 1).astype(np.float32))
            self._table, self._table_first, self._table_lr = table.to(self.scalars.device), t, group['lr']
        self.scalars.copy_(self._table[t - self._table_first], non_blocking=True)

    def step(self, users, queries, items, labels):
        if users.shape[0] != self.batch_rows:
            raise ValueError(f'this step was recorded for batches of {self.batch_rows} rows, got {users.shape[0]}')
        if self.stale():
            raise RuntimeError('CapturedTrainingStep: a setting baked into the recording changed (Adam betas / eps / weight_decay or batch_rows_only_last_layer - the values '
                               'every step() compares; the IHG_* environment and the switches of ihgnn_amd.ops are compared by stale(full=True), which the training loop '
                               'asks once per epoch); record a new step')
        self.users.copy_(users, non_blocking=True)
        self.queries.copy_(queries, non_blocking=True)
        self.items.copy_(items, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self._refresh_scalars()
        self.graph.replay()
        self.optimizer.advance()
        return self.loss
