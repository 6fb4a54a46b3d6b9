#!/usr/bin/env python3
"""N ranks of the HIP model == 1 rank on the union batch, runnable on ONE GPU.

    python tools/two_rank_check.py [--ranks 2] [--sync flat|bucketed|sharded|cotangent] [--device 0] [--backend gloo]

The parent starts the rank processes before touching the GPU (as bench.py does).  Every rank builds the same RawGnn replica on
GPU `--device` (RCCL refuses two ranks on one GPU, so the single-GPU form uses gloo; on a multi-GPU node pass `--backend nccl
--device -1`), trains two steps on its slice of one global batch with the chosen gradient exchange, and rank 0 then repeats the
two steps alone on the whole batch and compares every parameter.  Exit code 0 = equal within 2e-5 relative.

`--ranks 1 --backend nccl` is the single-GPU RCCL form: one rank in an RCCL process group with IHG_FORCE_COLLECTIVES=1, so every collective
of the chosen exchange (all-reduce with ReduceOp.AVG, async bucket launches from the gradient hooks, reduce_scatter_tensor,
all_gather_into_tensor, broadcast) really runs through RCCL - as identities - and the result must equal the plain one-rank training.
"""
import argparse
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ranks', type=int, default=2)
    ap.add_argument('--sync', default='bucketed')
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--backend', default='gloo')
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--collapsed', action='store_true', help='a graph with isolated nodes and repeated triples under IHG_COMPACT_NODES=1 / IHG_EDGE_MULTIPLICITY=1: the layout numbers '
                    'only the nodes that have hyperedges and keeps every distinct triple once (config C5\'s default) - the exchange maps the union\'s rows through the layout\'s node map')
    ap.add_argument('--steps', type=int, default=2, help='training steps; beyond a handful the comparison with the one-rank run is dropped (Adam amplifies rounding noise entry by entry) and only the '
                    'replicas are compared with each other: bitwise, under the cotangent exchange')
    ap.add_argument('--batch', type=int, default=64, help='batch rows per rank (8 ranks x 700: the union of the ranks\' 3 B batch rows exceeds 16,384 - the wide instance of the combine kernel)')
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ:
        import socket
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                 env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.ranks), MASTER_ADDR='127.0.0.1',
                                          MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0',
                                          **({'IHG_FORCE_COLLECTIVES': '1'} if args.ranks == 1 else {}))) for r in range(args.ranks)]
        codes = [k.wait() for k in kids]
        raise SystemExit(max(abs(c) for c in codes))

    import numpy as np
    import torch
    import torch.distributed as dist
    from ihgnn_amd import distributed as ihg_dist, synth
    from ihgnn_amd.Dataset import GraphDataset
    from ihgnn_amd.Models import HemPredictionLayer, IHGNNLayer, RawGnn
    from ihgnn_amd.optim import Adam

    rank, local, world = ihg_dist.init_from_env(args.backend)
    dev = torch.device(f'cuda:{args.device if args.device >= 0 else local}')
    torch.cuda.set_device(dev)
    w = synth.draw(300, 40, 200, 50, 4000, seed=21)
    triples = w.triples
    if args.collapsed:
        from ihgnn_amd import layout as layout_mod
        layout_mod.COMPACT_NODES = layout_mod.EDGE_MULTIPLICITY = '1'
        g = np.random.default_rng(4)
        live = [g.choice(n, n // 3, replace=False) for n in (300, 40, 200)]           # two thirds of every type never appear: isolated nodes, which the batches do sample
        base = np.stack([g.choice(live[k], 3000) for k in range(3)], 1)
        triples = np.concatenate([base, base[:1500]])
    ds = GraphDataset.from_arrays(w.user_count, w.query_count, w.item_count, w.vocab_size, w.bag_words, w.bag_offsets, triples, device=dev)
    if args.collapsed:
        lay = ds.hypergraph.layout
        assert lay.compact and lay.edge_weight is not None and lay.node_count == sum(len(x) for x in live)

    def replica():
        torch.manual_seed(5)
        return RawGnn(dev, ds, args.dim, IHGNNLayer, 2, 3, False, HemPredictionLayer, 0.5).to(dev)

    rng = np.random.default_rng(8)
    B = args.batch * world
    u, q, i = (torch.from_numpy(rng.integers(0, n, B)).to(dev) for n in (300, 40, 200))
    y = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(dev)

    model = replica()
    sync = ihg_dist.make_gradient_sync(model, args.sync)
    sync.broadcast_parameters(0)
    opt = sync.optimizer(1e-3) if sync.owns_optimizer else Adam(model.parameters(), 1e-3)
    rows = ihg_dist.shard_range(B, rank, world)
    sl = slice(rows.start, rows.stop)
    assert world > 1 or sync.distributed, 'a one-rank run must be forced through the collectives (IHG_FORCE_COLLECTIVES=1)'
    cotangent = getattr(sync, 'mode', None) == 'cotangent'  # the ranks exchange the batch rows' cotangents inside the backward; no dense gradient exchange
    for step in range(args.steps):
        (model.bce_loss(u[sl], q[sl], i[sl], y[sl], cotangent_sync=sync) if cotangent else model.bce_loss(u[sl], q[sl], i[sl], y[sl])).backward()
        if cotangent:
            assert sync.sent_bytes == 3 * (rows.stop - rows.start) * (8 + 4 * (args.dim * 3 + 4)), sync.sent_bytes
        sync.average_gradients()
        opt.step()
        sync.zero_grad()
        if step == 0 and sync.owns_optimizer:                # the sharded optimizer's checkpoint: gathered over ranks, loadable again
            state = opt.state_dict()
            assert sum(e['exp_avg'].numel() for e in state['state'].values()) == sync.flat.numel()       # (torch.optim.Adam's layout)
            opt.load_state_dict(state)
    torch.cuda.synchronize()
    ok = True
    if cotangent:
        drift = sync.check_replicas()                        # nothing but row cotangents crossed between the ranks: the replicas must be bitwise identical
        if drift != 0.0:
            ok = False
            print(f'REPLICAS DIVERGED under the cotangent exchange: {drift:.3e}', flush=True)
    if args.steps > 4:
        if not cotangent:                                    # the dense exchanges: the replicas hold the same averaged gradient, their parameters must agree bitwise as well
            worst = 0.0
            for p in model.parameters():
                ref = p.data.clone()
                dist.broadcast(ref, src=0)
                worst = max(worst, float((ref - p.data).abs().max()))
            if worst != 0.0:
                ok = False
                print(f'REPLICAS DIVERGED under the {args.sync} exchange: {worst:.3e}', flush=True)
        if rank == 0:
            print(f'two_rank_check: sync={args.sync} ranks={world} steps={args.steps}: replicas identical -> {"OK" if ok else "FAIL"}', flush=True)
    elif rank == 0:
        alone = replica()
        opt1 = Adam(alone.parameters(), 1e-3)
        for _ in range(2):
            alone.bce_loss(u, q, i, y).backward()
            opt1.step(); opt1.zero_grad()
        worst = 0.0
        for (name, a), (_, b) in zip(model.state_dict().items(), alone.state_dict().items()):
            err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            worst = max(worst, err)
            if err > 2e-5:
                ok = False
                print(f'MISMATCH {name}: {err:.3e}', flush=True)
        print(f'two_rank_check: sync={args.sync} ranks={world} worst relative deviation {worst:.3e} -> {"OK" if ok else "FAIL"}', flush=True)
    dist.barrier()
    dist.destroy_process_group()
    raise SystemExit(0 if ok else 1)


if __name__ == '__main__':
    main()
