#!/usr/bin/env python3
"""torch.profiler view of one training step at a bench.py workload: which ATen ops (and which of our ops) own the device time.

    python tools/step_trace.py [--config C2] [--steps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from ihgnn_amd import synth
from ihgnn_amd.Dataset import GraphDataset


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C2')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--layer', default='ihgnn')
    ap.add_argument('--order', type=int, default=3)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = synth.CONFIGS[args.config]
    w = synth.draw_config(args.config)
    ds = GraphDataset.from_arrays(w.user_count, w.query_count, w.item_count, w.vocab_size, w.bag_words, w.bag_offsets, w.triples, device=dev)
    model = bench.build_model(ds, dev, args.layer, cfg['layers'], args.order, cfg['dim'])
    from ihgnn_amd.optim import Adam
    opt = Adam(model.parameters(), 1e-3, weight_decay=0)
    lossf = torch.nn.BCEWithLogitsLoss()
    batches = list(ds.sample_batches(100, args.steps + 3, seed=1000))

    def step(k):
        u, q, i, y = batches[k]
        loss = model.bce_loss(u, q, i, y) if model.supports_fused_loss(lossf) else lossf(model(u, q, i), y)
        loss.backward()
        opt.step()
        opt.zero_grad()

    for k in range(3):
        step(k)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for k in range(3, 3 + args.steps):
            step(k)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=60,
                                                             max_shapes_column_width=70))


if __name__ == '__main__':
    main()
