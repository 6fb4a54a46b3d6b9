#!/usr/bin/env python3
"""Do the gather launches over the two-hop lists run faster in COLUMN PASSES?  A pass over a quarter of the columns gathers 128-byte pieces of the node rows: the
table slice a pass touches is a quarter of the table (C3: 48 MB instead of 192 MB), its hot rows (every query row: 26 k x 128 B = 3.3 MB) fit an XCD's 4 MB of L2,
and the id lists are read once per pass (8 B per incidence against 128 B gathered).  Same kernel (it takes row strides), same sums, bitwise.

    python tools/column_pass_probe.py [--config C3] [--rounds 6]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ihgnn_amd import ops, synth
from ihgnn_amd.layout import IncidenceLayout


def timed(fn, rounds):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C3')
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--dim', type=int, default=0)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = synth.CONFIGS[args.config]
    d = args.dim or cfg['dim']
    w = synth.draw_config(args.config)
    lay = IncidenceLayout(w.triples, w.user_count, w.query_count, w.item_count, dev)
    N = lay.node_count
    torch.manual_seed(0)
    x = torch.randn(N, d, device=dev)
    out = torch.empty(N, d, device=dev)
    ref = ops.node_segment_sum_raw(x, lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight).clone()
    print(f'{args.config}: N={N} E={lay.edge_count} d={d}')
    for parts in (1, 2, 4, 8):
        wd = d // parts
        if wd % 4 or (args.only and str(parts) not in args.only.split(',')):
            continue

        def two_hop():
            for c in range(parts):
                ops.node_segment_sum_raw(x[:, c * wd:(c + 1) * wd], lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight, out=out[:, c * wd:(c + 1) * wd])
        med, best = timed(two_hop, args.rounds)
        same = torch.equal(out, ref)
        sums = [torch.empty(N, 3 * wd, device=dev) for _ in range(parts)]

        def pair_sums():
            for c in range(parts):
                ops.node_pair_sums_raw(x[:, c * wd:(c + 1) * wd], lay, out=sums[c])
        med_p, best_p = timed(pair_sums, args.rounds)
        print(f'column passes {parts} (width {wd:3d}): two-hop {med:8.1f} us (best {best:8.1f}) bitwise {same} | pair sums {med_p:8.1f} us (best {best_p:8.1f})')


if __name__ == '__main__':
    main()
