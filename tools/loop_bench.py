#!/usr/bin/env python3
"""Steps per second of the reference-shaped training loop (DataLoader + collate_fn + TrainTestHelper) at a bench.py workload:
what `python -m ihgnn_amd.Main` achieves end to end, host-side sampling included.

    python tools/loop_bench.py [--config C2] [--steps 300]
"""
import argparse
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils.data import DataLoader

import bench
from ihgnn_amd import synth
from ihgnn_amd.Dataset import GraphDataset
from ihgnn_amd.optim import Adam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C2')
    ap.add_argument('--steps', type=int, default=300)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = synth.CONFIGS[args.config]
    w = synth.draw_config(args.config)
    ds = GraphDataset.from_arrays(w.user_count, w.query_count, w.item_count, w.vocab_size, w.bag_words, w.bag_offsets, w.triples, device=dev)
    model = bench.build_model(ds, dev, 'ihgnn', cfg['layers'], 3, cfg['dim'])
    opt = Adam(model.parameters(), 1e-3)
    lossf = torch.nn.BCEWithLogitsLoss()
    loader = DataLoader(ds, batch_size=100, shuffle=True, collate_fn=GraphDataset.collate_fn)

    def run(n):
        for p_u, p_q, p_i, p_f, n_u, n_q, n_i, n_f in itertools.islice(loader, n):
            users, queries, items = torch.cat([p_u, n_u]), torch.cat([p_q, n_q]), torch.cat([p_i, n_i])
            loss = model.bce_loss(users, queries, items, torch.cat([p_f, n_f]).float())
            loss.backward()
            opt.step()
            opt.zero_grad()

    run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    t1 = time.perf_counter()
    for _ in itertools.islice(loader, args.steps):
        pass
    host = (time.perf_counter() - t1) / args.steps
    print(f'{args.config}: {1e3 * dt:.3f} ms per step through DataLoader + collate_fn ({1e3 * host:.3f} ms of it is host-side sampling + H2D when run alone)')


if __name__ == '__main__':
    main()
