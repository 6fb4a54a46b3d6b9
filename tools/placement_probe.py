#!/usr/bin/env python3
"""Does the PLACEMENT of a gathered table matter?  The two-hop launch (K7 over hop2_csr) of config C5 on source tables that differ only in where the allocator put them:
a fresh segment of its own, the head / the middle of a large cached block, odd offsets inside one.  (Round 5: in a recorded step the last layer's two-hop forward ran
13.7 ms against 22.6 ms in the eager step - same kernel, same arguments but the addresses.)

    python tools/placement_probe.py [--config C5] [--scale 1.0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ihgnn_amd import ops, synth
from ihgnn_amd.layout import IncidenceLayout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C5')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--rounds', type=int, default=3)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = synth.CONFIGS[args.config]
    d = cfg['dim']
    w = synth.draw_config(args.config, scale=args.scale)
    lay = IncidenceLayout(w.triples, w.user_count, w.query_count, w.item_count, dev)
    del w
    N = lay.node_count
    print(f'{args.config}: N={N} E={lay.edge_count} d={d}: table {N * d * 4 / 1e9:.2f} GB')

    def timed(x, out, label):
        ts = []
        for r in range(args.rounds + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.node_segment_sum_raw(x, lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight, role='k7.two_hop', out=out)
            b.record()
            torch.cuda.synchronize()
            if r:
                ts.append(a.elapsed_time(b))
        p, q = x.data_ptr(), out.data_ptr()
        print(f'{label:58s} min {min(ts):8.3f} ms  avg {sum(ts) / len(ts):8.3f} | src @ {p:#x} (mod 2 MiB {p % (2 << 20):#x}, mod 1 GiB {p % (1 << 30):#x}) out @ {q:#x}')

    def fill(x):
        step = 1 << 20
        for lo in range(0, x.shape[0], step):
            x[lo:lo + step].normal_()
        return x

    def stats(tag):
        s = torch.cuda.memory_stats()
        print(f'   [{tag}] segments {s["segment.all.current"]}, reserved {s["reserved_bytes.all.current"] / 1e9:.1f} GB, allocated {s["allocated_bytes.all.current"] / 1e9:.1f} GB')

    torch.cuda.empty_cache()
    stats('start')
    x = fill(torch.empty(N, d, device=dev))
    out = torch.empty(N, d, device=dev)
    timed(x, out, 'fresh segments of their own (table, output)')
    out2 = torch.empty(N, d, device=dev)
    timed(x, out2, 'the same table, another fresh output')
    del out2
    # the same launch replayed from a recording (hipGraph)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.node_segment_sum_raw(x, lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight, role='k7.two_hop', out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.node_segment_sum_raw(x, lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight, role='k7.two_hop', out=out)
        inside = torch.empty(N, d, device=dev)                             # a table that lives in the recording's own pool
        inside.copy_(x)
        out_inside = ops.node_segment_sum_raw(inside, lay.hop2_csr, None, lay.inv_deg, 1, self_weight=lay.self_weight, role='k7.two_hop')
    ts = []
    for r in range(args.rounds + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    print(f'{"recording: [launch on the first table; copy into the pool; launch on that copy]":58s} min {min(ts):8.3f} ms  (the copy moves 20.5 GB: ~ 5 ms) | pool table @ {inside.data_ptr():#x}')
    timed(inside, out, 'eager launch on the recording\'s pool table')
    del graph, inside, out_inside
    # a large block, freed into the allocator's cache, then the table carved out of it
    big = torch.empty(int(100e9) // 4, device=dev)
    base = big.data_ptr()
    del big
    stats('100 GB block cached')
    pad = torch.empty(int(7.3e9) // 4, device=dev)                          # the head of the cached block
    y = fill(torch.empty(N, d, device=dev))                                  # ... the table behind it
    print(f'   table at offset {(y.data_ptr() - base) / 1e9:.3f} GB of the cached block')
    timed(y, out, 'carved out of a cached 100 GB block (behind 7.3 GB)')
    y.copy_(x)
    timed(y, out, '... with the first table\'s values')
    del pad, y
    big = torch.empty(int(100e9) // 4, device=dev)
    for off_bytes in (0, 4096, (1 << 20) + 4096, (1 << 30) + (3 << 20)):
        v = big[off_bytes // 4: off_bytes // 4 + N * d].view(N, d)
        v.copy_(x)
        timed(v, out, f'view at byte offset {off_bytes} of a 100 GB tensor')
    timed(x, big[int(60e9) // 4: int(60e9) // 4 + N * d].view(N, d), 'first table, output inside the 100 GB tensor')
    del big
    torch.cuda.empty_cache()
    stats('cache emptied')
    z = torch.empty(N, d, device=dev)
    z.copy_(x)
    timed(z, out, 'a fresh segment again')


if __name__ == '__main__':
    main()
