#!/usr/bin/env python3
"""Per-step wall time and allocator statistics of the first training steps at a bench config (is the first timed step after one warm-up step honest?).

    python tools/c5_steps.py [--config C5] [--steps 5]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ihgnn_amd import ops, synth
from ihgnn_amd.Dataset import GraphDataset
from ihgnn_amd.optim import Adam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C5')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--late-adam-state', action='store_true', help="create Adam's moment buffers in the first optimizer step (torch's default)")
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = synth.CONFIGS[args.config]
    w = synth.draw_config(args.config)
    ds = GraphDataset.from_arrays(w.user_count, w.query_count, w.item_count, w.vocab_size, w.bag_words, w.bag_offsets, w.triples, device=dev)
    model = bench.build_model(ds, dev, 'ihgnn', cfg['layers'], 3, cfg['dim'])
    model.batch_rows_only_last_layer = False
    opt = Adam(model.parameters(), 1e-3, weight_decay=0)
    if '--late-adam-state' not in sys.argv:
        opt.ensure_state()                                   # what bench.py does: the moment buffers exist before the first step
    batches = list(ds.sample_batches(100, args.steps, seed=1000))
    for k, (u, q, i, y) in enumerate(batches):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = model.bce_loss(u, q, i, y)
        ops.backward(loss)
        opt.step()
        opt.zero_grad()
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats()
        print(f'step {k}: {1e3 * (time.perf_counter() - t0):9.1f} ms   reserved {st["reserved_bytes.all.current"] / 2**30:7.1f} GiB  allocated peak {st["allocated_bytes.all.peak"] / 2**30:7.1f} GiB  '
              f'device mallocs so far {st["num_device_alloc"]}  frees {st["num_device_free"]}', flush=True)


if __name__ == '__main__':
    main()
