"""``Adam``: ``torch.optim.Adam`` for the parameters of this package's models, stepped by one HIP launch.

Same constructor keywords, update rule, ``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter) and
checkpoint compatibility as ``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (reference ``Main.py:192``; amsgrad,
maximize and tensor learning rates are not carried).  The elementwise update is a pure HBM stream; torch's multi-tensor kernel
reaches ~3.2 TB/s on the 16 M embedding parameters of the C2 workload, ``ihg_adam_step`` is a single vectorised pass.
"""
import ctypes
from typing import Iterable

import torch
from torch.optim import Optimizer

from . import _lib


class _AdamTensor(ctypes.Structure):
    _fields_ = [('param', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('exp_avg', ctypes.c_void_p), ('exp_avg_sq', ctypes.c_void_p),
                ('count', ctypes.c_int64)]


class Adam(Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError(f'invalid Adam hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                    raise _lib.IhgnnHipError('ihgnn_amd.optim.Adam steps dense float32 GPU parameters (no CPU path)')
                if not p.is_contiguous():
                    raise _lib.IhgnnHipError('ihgnn_amd.optim.Adam needs contiguous parameters')
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = torch.tensor(0.0)                      # host-side counter, as torch keeps it for non-capturable Adam
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if torch.is_tensor(state['step']) and state['step'].is_cuda:   # a checkpoint loaded with map_location=device: keep the
                    state['step'] = state['step'].cpu()                        # counter on the host (int() below would sync every step)
                state['step'] += 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(int(state['step']), []).append((p, grad, state))
            beta1, beta2 = group['betas']
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for step, items in by_step.items():
                table = (_AdamTensor * len(items))()
                for slot, (p, grad, state) in zip(table, items):
                    slot.param, slot.grad = p.data_ptr(), grad.data_ptr()
                    slot.exp_avg, slot.exp_avg_sq = state['exp_avg'].data_ptr(), state['exp_avg_sq'].data_ptr()
                    slot.count = p.numel()
                _lib.check(lib.ihg_adam_step(ctypes.cast(table, ctypes.c_void_p), len(items), float(group['lr']), float(beta1), float(beta2),
                                             float(group['eps']), float(group['weight_decay']), step, stream), 'ihg_adam_step')
        return loss

    # ------------------------------------------------------------------ recorded steps (ihgnn_amd/captured_step.py)
    def next_step(self, group_index: int = 0) -> int:
        """The step number the next update of a parameter group will carry (all its parameters share one count)."""
        group = self.param_groups[group_index]
        steps = {int(self.state[p]['step']) if len(self.state.get(p, {})) else 0 for p in group['params'] if p.requires_grad}
        if len(steps) > 1:
            raise _lib.IhgnnHipError('a recorded Adam step needs one step count per parameter group')
        return (steps.pop() if steps else 0) + 1

    @torch.no_grad()
    def ensure_state(self) -> None:
        """Create the moment buffers of every trainable parameter now (a recording must not allocate them)."""
        for group in self.param_groups:
            for p in group['params']:
                if p.requires_grad and len(self.state[p]) == 0:
                    self.state[p]['step'] = torch.tensor(0.0)
                    self.state[p]['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    self.state[p]['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)

    @torch.no_grad()
    def launch_with_device_scalars(self, scalars: torch.Tensor) -> None:
        """The update of parameter group 0 with the two step-dependent scalars taken from ``scalars`` (float32 ``[2]`` on the device) when the
        kernel runs; the host-side step counters are NOT advanced (``advance()`` does that once per replay)."""
        lib = _lib.load()
        if len(self.param_groups) != 1:
            raise _lib.IhgnnHipError('a recorded Adam step handles one parameter group')
        group = self.param_groups[0]
        items = []
        for p in group['params']:
            if p.grad is None:
                continue
            state = self.state[p]
            if len(state) == 0:
                state['step'] = torch.tensor(0.0)
                state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not p.grad.is_contiguous() or not p.is_contiguous():
                raise _lib.IhgnnHipError('a recorded Adam step needs contiguous parameters and gradients')
            items.append((p, p.grad, state))
        table = (_AdamTensor * len(items))()
        for slot, (p, grad, state) in zip(table, items):
            slot.param, slot.grad = p.data_ptr(), grad.data_ptr()
            slot.exp_avg, slot.exp_avg_sq = state['exp_avg'].data_ptr(), state['exp_avg_sq'].data_ptr()
            slot.count = p.numel()
        beta1, beta2 = group['betas']
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.ihg_adam_step_device_scalars(ctypes.cast(table, ctypes.c_void_p), len(items), float(beta1), float(beta2), float(group['eps']),
                                                    float(group['weight_decay']), ctypes.c_void_p(scalars.data_ptr()), stream), 'ihg_adam_step_device_scalars')

    def advance(self) -> None:
        """Count one step on every parameter (after a replay of a recorded step)."""
        for group in self.param_groups:
            for p in group['params']:
                state = self.state.get(p)
                if state:
                    if torch.is_tensor(state['step']) and state['step'].is_cuda:
                        state['step'] = state['step'].cpu()
                    state['step'] += 1
