#!/usr/bin/env python3
"""What the user rows cost in K7's hyperedge -> node launch (the bound on what an on-chip user-run reduction in the producer could save):
the launch over all rows against the launch over the query and item rows only, at a bench workload.  python tools/k7_user_share.py [--config C3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ihgnn_amd import ops, profiler, synth
from ihgnn_amd.layout import IncidenceLayout

ap = argparse.ArgumentParser(); ap.add_argument('--config', default='C3'); args = ap.parse_args()
dev = torch.device('cuda:0')
cfg = synth.CONFIGS[args.config]; w = synth.draw_config(args.config)
lay = IncidenceLayout(w.triples, w.user_count, w.query_count, w.item_count, dev)
d = cfg['dim']
ef = torch.randn(lay.edge_count, d, device=dev)
lens = np.diff(lay.node_csr.ptr_host.astype(np.int64))
u = lay.user_count
qi = (u + np.argsort(-lens[u:], kind='stable')).astype(np.int32)
qi_rows = torch.from_numpy(qi).to(dev)
out = torch.empty(lay.node_count, d, device=dev)
for _ in range(3):
    ops.node_segment_sum_raw(ef, lay.node_csr, None, lay.inv_deg, 1, out=out, role='all_rows')
    ops.node_segment_sum_raw(ef, lay.node_csr, None, lay.inv_deg, 1, out=out, rows=qi_rows, role='query_item_rows')
profiler.start()
for _ in range(10):
    ops.node_segment_sum_raw(ef, lay.node_csr, None, lay.inv_deg, 1, out=out, role='all_rows')
    ops.node_segment_sum_raw(ef, lay.node_csr, None, lay.inv_deg, 1, out=out, rows=qi_rows, role='query_item_rows')
profiler.stop()
for k, v in profiler.summary().items():
    print(f'{args.config} {k:24s} avg {v["avg_us"]:8.1f} us')
