#!/usr/bin/env python3
"""Does the ORDER in which the gather launches walk their destination rows matter?  

The two-hop and pair-sum launches gather 6 E rows of the node table per launch at a 27 % L2 hit rate.  The layout owns the order of the work list
(`Csr.row_order`: rows by decreasing length) and, through it, which XCD's L2 a row's gathers go through: unit u of the list is taken by workgroup
(u / 8) mod grid, and workgroups are dealt to the 8 XCDs round-robin, so units with equal (u / 8) mod 8 share one 4 MB L2.  This probe re-orders the work list -
per-row sums keep their order, results are bitwise the same - and times the launches; run under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` for the hit rates.

    python tools/locality_probe.py --config C3 [--variants base,key,xcd,random] [--rounds 6]

variants   base    rows by decreasing length (what the layout builds)
           key     ... and, among rows of equal length, by a shared-neighbour key: the most frequent OTHER member of the row's hyperedges (users, items: their
                   dominant query; queries: their dominant item)
           xcd     rows cut into 8 groups by that key (equal total list length per group), each group by decreasing length, group g placed on the units
                   that XCD g takes: rows that gather the same sources meet in one L2
           random  rows of equal length in random order (control: how much the order matters at all)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ihgnn_amd import ops, profiler, synth
from ihgnn_amd.layout import IncidenceLayout


def dominant_neighbour(lay):
    """Per node: the most frequent id among a chosen other-member slot of its hyperedges (queries for users and items, items for queries); -1 for isolated nodes."""
    ptr = lay.hop2_csr.ptr_host.astype(np.int64)
    ids = lay.hop2_csr.ids_host.astype(np.int64).reshape(-1, 2)         # (a, b) per incidence
    n, u, q = lay.node_count, lay.user_count, lay.query_count
    rows = np.repeat(np.arange(n), np.diff(ptr) // 2)
    # users: a = query; queries: b = item; items: b = query
    pick = np.where(rows < u, ids[:, 0], ids[:, 1])
    order = np.lexsort((pick, rows))
    r, p = rows[order], pick[order]
    start = np.r_[True, (r[1:] != r[:-1]) | (p[1:] != p[:-1])]
    run_id = np.cumsum(start) - 1
    run_len = np.bincount(run_id)
    run_row, run_pick = r[start], p[start]
    best = np.full(n, -1, np.int64)
    best_len = np.zeros(n, np.int64)
    o = np.argsort(run_len, kind='stable')                               # later (longer) runs overwrite shorter ones
    best[run_row[o]] = run_pick[o]
    return best


def make_order(lay, variant, seed=0):
    csr = lay.hop2_csr
    lens = np.diff(csr.ptr_host.astype(np.int64))
    key_len = lens.copy()
    key_len[lens > csr.heavy_threshold] = -1
    if variant == 'base':
        return np.argsort(-key_len, kind='stable').astype(np.int32)
    if variant == 'random':
        rnd = np.random.default_rng(seed).random(lens.shape[0])
        return np.lexsort((rnd, -key_len)).astype(np.int32)
    dom = dominant_neighbour(lay)
    if variant == 'key':
        return np.lexsort((dom, -key_len)).astype(np.int32)
    if variant == 'xcd':
        by_key = np.argsort(dom, kind='stable')
        work = np.cumsum(np.maximum(key_len[by_key], 0) + 8)             # (+8: the per-row overhead, so that empty rows are spread too)
        group = np.minimum((work * 8 // (work[-1] + 1)), 7)
        n = lens.shape[0]
        out = np.full(((n + 63) // 64 + 8) * 64, -1, np.int64)
        # group g, position p of its length-sorted list -> unit ((p // 8) * 8 + g) * 8 + p % 8
        for g in range(8):
            rows = by_key[group == g]
            rows = rows[np.argsort(-key_len[rows], kind='stable')]
            p = np.arange(rows.shape[0])
            pos = ((p // 8) * 8 + g) * 8 + p % 8
            if pos.max(initial=0) >= out.shape[0]:
                out = np.concatenate([out, np.full(pos.max() + 64 - out.shape[0], -1, np.int64)])
            out[pos] = rows
        out = out[out >= 0]                                              # (the groups differ a little in row count: holes close up, the tail of the list loses its placement)
        assert out.shape[0] == n and np.unique(out).shape[0] == n
        return out.astype(np.int32)
    raise ValueError(variant)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C3')
    ap.add_argument('--variants', default='base,key,xcd,random')
    ap.add_argument('--rounds', type=int, default=6)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = synth.CONFIGS[args.config]
    d = cfg['dim']
    w = synth.draw_config(args.config)
    lay = IncidenceLayout(w.triples, w.user_count, w.query_count, w.item_count, dev)
    torch.manual_seed(0)
    x = torch.randn(lay.node_count, d, device=dev)
    base_order = lay.hop2_csr.row_order
    want_two = want_pairs = None
    for variant in args.variants.split(','):
        order = torch.from_numpy(make_order(lay, variant)).to(dev)
        lay.hop2_csr.row_order = order
        for _ in range(2):
            two = ops.node_segment_sum_raw(x, lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight, role='k7.two_hop')
            pairs = ops.node_pair_sums_raw(x, lay)
        if want_two is None:
            want_two, want_pairs = two.clone(), pairs.clone()
        same = bool(torch.equal(two, want_two) and torch.equal(pairs, want_pairs))           # per-row sums keep their order: bitwise the same
        torch.cuda.synchronize()
        profiler.start()
        for _ in range(args.rounds):
            ops.node_segment_sum_raw(x, lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight, role='k7.two_hop')
            ops.node_pair_sums_raw(x, lay)
        profiler.stop()
        s = profiler.summary()
        print(f'{args.config} {variant:7s} two_hop {s["k7.two_hop"]["avg_us"]:8.1f} us (min {1e3 * min(s["k7.two_hop"]["times_ms"]):8.1f})   pair_sums {s["node_pair_sums"]["avg_us"]:8.1f} us '
              f'(min {1e3 * min(s["node_pair_sums"]["times_ms"]):8.1f})   bitwise_same_as_base={same}', flush=True)
    lay.hop2_csr.row_order = base_order


if __name__ == '__main__':
    main()
