#!/usr/bin/env python3
"""Micro-benchmark of the individual HIP ops at a bench.py workload shape (interleaved rounds, HIP-event timing).

    python tools/kbench.py [--config C2] [--rounds 10] [--ops k5,k7,pairs,twohop,interact,layer0,layer,ifwd,linear]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ihgnn_amd import ops, profiler, synth
from ihgnn_amd.layout import IncidenceLayout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C2')
    ap.add_argument('--rounds', type=int, default=10)
    ap.add_argument('--dim', type=int, default=0)
    ap.add_argument('--order', type=int, default=3)
    ap.add_argument('--ops', default='k5,k7,interact,linear')
    ap.add_argument('--scale', type=float, default=1.0)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = synth.CONFIGS[args.config]
    d = args.dim or cfg['dim']
    w = synth.draw_config(args.config, scale=args.scale)
    lay = IncidenceLayout(w.triples, w.user_count, w.query_count, w.item_count, dev)
    E, N = lay.edge_count, lay.node_count
    print(f'{args.config}: N={N} E={E} d={d} heavy_rows={lay.node_csr.n_heavy} segments={lay.node_csr.n_segments}')
    torch.manual_seed(0)
    x = torch.randn(N, d, device=dev)
    dy_layer = torch.randn(N, d, device=dev)
    ef = torch.randn(E, d, device=dev)
    k = 7 if args.order == 3 else 6
    wa = (torch.randn(d, k * d, device=dev) / (k * d) ** 0.5).requires_grad_(True)
    wt = (torch.randn(d, d, device=dev) / d ** 0.5).requires_grad_(True)
    b = torch.randn(d, device=dev, requires_grad=True)
    want = set(args.ops.split(','))
    listed = torch.randint(0, N, (3300,), device=dev)
    mask = torch.zeros(N, dtype=torch.uint8, device=dev)
    mask[listed] = 1
    xm = x * mask[:, None].float()

    def run_once():
        if 'k5' in want:
            ops.edge_gather_sum_raw(x, lay.i3)
            ops.edge_gather_sum_raw(x, lay.i3, lay.inv_deg)
        if 'k7' in want:
            ops.node_segment_sum_raw(ef, lay.node_csr, None, lay.inv_deg, 1)
        if 'pairs' in want:                                               # pair sums of the interactive layer's node-level form
            ops.node_pair_sums_raw(x, lay)
        if 'masked' in want:                                              # the last layer's backward: the two-hop pull with 3,300 listed source rows
            ops.node_segment_sum_raw(xm, lay.hop2_csr, lay.inv_deg, None, 0, self_weight=lay.self_weight, src_mask=mask, role='k7.two_hop_bwd_masked')
        if 'twohop' in want:                                              # a first-order layer's two-hop launch
            ops.node_segment_sum_raw(x, lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight, role='k7.two_hop')
        if 'linear' in want:
            xr = x.detach().requires_grad_(True)
            y = ops.node_linear(xr, wt, b, lay)
            y2 = ops.node_linear(y, wa, b, lay, typed=True, bias_mask=1)
            y2.backward(x)
        if 'layer0' in want:                                              # interactive step + hyperedge -> node pass as one autograd node (the gathering backward at d = 128)
            hr = x.detach().requires_grad_(True)
            pr = x.detach().requires_grad_(True)
            ops.interact_to_nodes(hr, pr, wa, lay, args.order, lay.inv_deg).backward(x)
        if 'layer' in want:                                               # the whole interactive layer as one autograd node (what IHGNNLayer runs)
            hr = x.detach().requires_grad_(True)
            ops.interact_layer(hr, wa, b, lay, args.order, lay.inv_deg).backward(dy_layer)       # (a cotangent of its own: with `x` for both, the member-gradient
                                                                                                 # kernel gathers h and dy rows out of ONE table - 8 % faster than in a step)
        if 'ifwd' in want:                                                # the interactive step's forward alone
            with torch.no_grad():
                ops.interact(x, x, wa, lay, args.order)
        if 'interact' in want:
            hr = x.detach().requires_grad_(True)
            pr = x.detach().requires_grad_(True)
            out = ops.interact(hr, pr, wa, lay, args.order)
            out.backward(ef)

    for _ in range(2):
        run_once()
    torch.cuda.synchronize()
    profiler.start()
    for _ in range(args.rounds):
        run_once()
    profiler.stop()
    for name, v in sorted(profiler.summary().items()):
        ts = sorted(v['times_ms'])
        print(f'{name:28s} launches {v["launches"]:4d}  avg {v["avg_us"]:9.1f} us  min {1e3 * ts[0]:9.1f}  med {1e3 * ts[len(ts) // 2]:9.1f}')
    hbm = {'edge_gather_sum': E * (16 * d + 12), 'node_segment_sum': E * (12 * d + 12) + N * (4 * d + 8)}
    for name, nbytes in hbm.items():
        if name in profiler.summary():
            print(f'  {name}: algorithmic {nbytes / 1e9:.3f} GB -> {nbytes / (profiler.summary()[name]["avg_us"] * 1e-6) / 1e12:.2f} TB/s')


if __name__ == '__main__':
    main()
