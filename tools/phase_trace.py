#!/usr/bin/env python3
"""Where a phase of the member-gradient kernel goes: clock stamps taken inside the kernel by an ablation build with -DIHG_ABL_M_TRACE (csrc/split_arith.hip: lane 0 of
matrix wave 0 and of service wave 4 of workgroup 40, phases 200 .. 1,223).

    bash tools/ab_variant.sh trace -DIHG_ABL_M_TRACE [more switches]
    IHG_ALLOW_ABLATION_BUILD=1 IHGNN_HIP_LIBRARY=build_ab/lib_trace.so python tools/phase_trace.py [--config C5 --scale 0.2 --op layer]

Prints, averaged over the stamped phases and in clock ticks: the phase length, and per role the time from the phase's start (its release from the barrier) to each mark."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ihgnn_amd import _lib, ops, synth
from ihgnn_amd.layout import IncidenceLayout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C5')
    ap.add_argument('--scale', type=float, default=0.2)
    ap.add_argument('--op', default='layer', choices=['layer', 'interact'])
    ap.add_argument('--order', type=int, default=3)
    ap.add_argument('--kernel', default='members', choices=['members', 'linear'], help='linear: the node-level linear backward (-DIHG_ABL_D_TRACE)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    d = synth.CONFIGS[args.config]['dim']
    w = synth.draw_config(args.config, scale=args.scale)
    lay = IncidenceLayout(w.triples, w.user_count, w.query_count, w.item_count, dev)
    N, E = lay.node_count, lay.edge_count
    torch.manual_seed(0)
    x, dy = torch.randn(N, d, device=dev), torch.randn(N, d, device=dev)
    k = 7 if args.order == 3 else 6
    wa = (torch.randn(d, k * d, device=dev) / (k * d) ** 0.5).requires_grad_(True)
    b = torch.randn(d, device=dev, requires_grad=True)
    for _ in range(2):
        hr = x.detach().requires_grad_(True)
        if args.op == 'layer':
            ops.interact_layer(hr, wa, b, lay, args.order, lay.inv_deg).backward(dy)
        else:
            ops.interact(hr, x, wa, lay, args.order).backward(torch.randn(E, d, device=dev))
    torch.cuda.synchronize()
    lib = _lib.load()
    if args.kernel == 'linear':
        wt = (torch.randn(d, d, device=dev) / d ** 0.5).requires_grad_(True)
        for _ in range(2):
            xr = x.detach().requires_grad_(True)
            ops.node_linear(xr, wt, b, lay).backward(dy)
        torch.cuda.synchronize()
        fn = lib.ihg_ablation_trace_dense
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p]
        out = np.zeros((64, 6), np.uint64)
        assert fn(out.ctypes.data) == 0
        t = out.astype(np.int64)
        n_ok = int(np.argmin((t > 0).all(1))) if not (t > 0).all() else 64
        t = t[2:n_ok - 1]
        print(f'{args.config} x {args.scale}: N = {N}, d = {d}; node-level linear backward, wave 0 of workgroup 40, {len(t)} phases, clock ticks from the phase\'s start')
        names = ['rows requested', 'dW contraction + next tile split', 'dx contraction', 'rows delivered + exponents published', 'dx stored (barrier reached)']
        for i, name in enumerate(names):
            print(f'  {name:44s} {np.mean(t[:, i + 1] - t[:, 0]):9.1f}')
        print(f'  phase length                                 {np.mean(np.diff(t[:, 0])):9.1f}')
        return
    fn = lib.ihg_ablation_trace
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    out = np.zeros((2, 1024, 4), np.uint64)
    assert fn(out.ctypes.data) == 0
    t = out.astype(np.int64)
    ok = (t[0] > 0).all(1) | (t[1] > 0).all(1)
    ok &= (t[0][:, :3] > 0).all(1) & (t[1] > 0).all(1)                  # (a workgroup with fewer phases than the stamped window leaves zeros)
    n_ok = int(np.argmin(ok)) if not ok.all() else len(ok)
    if n_ok < 8:
        sys.exit(f'only {n_ok} stamped phases: workgroup 40 has fewer than 208 phases at this size')
    t = t[:, :n_ok]
    m, s = t[0], t[1]
    phase = np.diff(m[:, 0])
    print(f'{args.config} x {args.scale}: E = {E}, d = {d}; {len(phase)} phases, clock ticks (clock64)')
    print(f'phase length            mean {phase.mean():9.1f}  median {np.median(phase):9.1f}  p90 {np.percentile(phase, 90):9.1f}')
    print(f'matrix wave 0:  MFMA loop done at {np.mean(m[:, 1] - m[:, 0]):9.1f}   run sums done (barrier reached) at {np.mean(m[:, 2] - m[:, 0]):9.1f}')
    print(f'service wave 4: requests issued at {np.mean(s[:, 1] - s[:, 0]):9.1f}   images written at {np.mean(s[:, 2] - s[:, 0]):9.1f}   epilogue done (barrier reached) at '
          f'{np.mean(s[:, 3] - s[:, 0]):9.1f}')
    print(f'service start - matrix start {np.mean(s[:, 0] - m[:, 0]):9.1f}')
    wait_m = m[1:, 0] - m[:-1, 2]
    wait_s = s[1:, 0] - s[:-1, 3]
    print(f'waiting at the barrier: matrix {wait_m.mean():9.1f}  service {wait_s.mean():9.1f}')


if __name__ == '__main__':
    main()
