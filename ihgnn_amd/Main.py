"""Training / evaluation driver: ``python -m ihgnn_amd.Main --ds <dataset> [...]``.

Same flags, defaults, call sequence, result-directory naming, checkpoint and metrics files as the reference's
``Main.py`` (lines 20-325), restricted to the RawGnn model with IHGNN / HGCN layers, and written as a function
so tests can drive it.  Launched under ``torchrun`` it trains data-parallel (one process per GPU, RCCL).
"""
import os
import sys
import time
from typing import Optional, Sequence

import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from . import distributed as ihg_dist
from .Dataset import GraphDataset, TestSearchLogDataLoader
from .Helpers.ArgsParser import parse_args
from .Helpers.GlobalSettings import Gs
from .Helpers.Graph import Pps2DGraph, PpsHyperGraph
from .Helpers.IOHelper import IOHelper
from .Helpers.Metrics import Metrics, MetricsCollection
from .Helpers.ProcessController import ProcessController
from .Helpers.TrainTestHelper import print_network_parameters, test_and_get_avg_metrics, train_and_get_avg_loss
from .Models import GCNLayer, HGCNLayer, HemPredictionLayer, IHGNNLayer, RawGnn, parse_gnn_layer, parse_model_type

DEFAULT_DATASET = 'AlibabaAir/Complete5Core/'


def result_directory(dataset_name: str, layers: int, layer_type: type, order: int, emb: int) -> str:
    parts = dataset_name.strip('/').split('/') + ['RawGnn', f'{layers}{layer_type.__name__}']
    if layer_type is IHGNNLayer:
        parts.append(f'O{order}')
    parts.append(f'emb{emb}')
    return os.path.join('Results', '-'.join(parts))


def main(argv: Optional[Sequence[str]] = None) -> MetricsCollection:
    args = parse_args(argv)
    rank, local_rank, world = ihg_dist.init_from_env()
    chief = rank == 0

    if getattr(args, 'seed', -1) is not None and int(getattr(args, 'seed', -1)) >= 0:
        # not in the reference (which seeds nothing): reproducible runs - the initial weights, the loader's shuffles and the sampled negatives follow from --seed
        import random
        import numpy
        random.seed(int(args.seed)); numpy.random.seed(int(args.seed)); torch.manual_seed(int(args.seed))
    Gs.graph_completeness = args.completeness
    Gs.long_tail_stat_fn = args.long_tail_filename or None
    Gs.embedding_size = args.embedding_size or Gs.embedding_size
    epoch_count = args.epoch_count or 110
    start_test = args.epoch_start_test or 10
    test_every = args.epoch_test_frequency or start_test
    dataset_name = args.dataset or DEFAULT_DATASET
    if (parse_model_type[args.model] or RawGnn) is not RawGnn:
        raise NotImplementedError('only the RawGnn model is part of this build')
    layer_type = parse_gnn_layer[args.gnn] or IHGNNLayer
    if layer_type not in (IHGNNLayer, HGCNLayer, GCNLayer):
        raise NotImplementedError(f'{layer_type.__name__} is outside the MI355X hypergraph path')
    layer_count = args.gnns or 2
    order = args.feature_order or 3
    if args.device == 'cpu':
        raise RuntimeError('ihgnn_amd has no CPU path: the hypergraph kernels are HIP-only')
    device = torch.device(f'cuda:{args.device}' if args.device else f'cuda:{local_rank}')
    torch.cuda.set_device(device)

    root = IOHelper.GetFirstLineContent('./dataset_dir.txt') if os.path.exists('./dataset_dir.txt') else './Data/'
    data_dir = os.path.join(root, dataset_name)
    result_dir = result_directory(dataset_name, layer_count, layer_type, order, Gs.embedding_size)
    os.makedirs(result_dir, exist_ok=True)
    stamp = time.strftime('%y%m%d-%H%M%S', time.localtime())
    fn_metrics = os.path.join(result_dir, f'{stamp}_metrics.txt')
    if chief:
        IOHelper.StartLogging(os.path.join(result_dir, f'{stamp}_train_log.txt' if args.storemetrics else 'train_log.txt'))
    else:
        IOHelper.StartLogging(None)
        IOHelper.quiet = True
    say = IOHelper.LogPrint if chief else (lambda *a, **k: None)

    say(f'device {device} | ranks {world} | batch {Gs.batch_size} | lr {Gs.learning_rate} | emb {Gs.embedding_size} | '
        f'L2 {Gs.weight_decay} | negatives {Gs.random_negative_sample_size}/{Gs.non_random_negative_sample_size}')
    say(f'model RawGnn | dataset {dataset_name} | {layer_count} x {layer_type.__name__} | order {order} | '
        f'query transform {Gs.Query.transform} | validation {Gs.use_valid_dataset}')
    say(f'store metrics {args.storemetrics} | store checkpoint {args.storecheckpoint} | load {args.checkpoint or False}\n')

    dataset_train = GraphDataset(
        fn_graph_info=os.path.join(data_dir, 'graph_info.txt'),
        fn_queries_multihot=os.path.join(data_dir, 'queries_multihot.txt'),
        fn_train_data=os.path.join(data_dir, 'train_data.csv'),
        graph_type=Pps2DGraph if layer_type is GCNLayer else PpsHyperGraph,
        random_negative_sample_size=Gs.random_negative_sample_size,
        non_random_negative_sample_size=Gs.non_random_negative_sample_size,
        device=device)
    if args.device_sampling:
        from .Dataset import DeviceBatchLoader
        dataloader_train = DeviceBatchLoader(dataset_train, Gs.batch_size, rank, world)
    elif world > 1:
        # data parallel: the ranks split every epoch's permutation between them (same number of steps on every rank, each row once
        # per epoch) and draw their negatives from differently seeded generators; global batch = batch_size x ranks
        import random
        random.seed(0x5eed + 7919 * rank)
        sampler = ihg_dist.ShardedBatchSampler(len(dataset_train), Gs.batch_size, rank, world, shuffle=True, seed=0)
        dataloader_train = DataLoader(dataset_train, batch_sampler=sampler, collate_fn=GraphDataset.collate_fn)
    else:
        dataloader_train = DataLoader(dataset_train, Gs.batch_size, shuffle=True, collate_fn=GraphDataset.collate_fn)
    dataloader_valid = TestSearchLogDataLoader(os.path.join(data_dir, 'valid_data.csv'), dataset_train, device)
    dataloader_test = TestSearchLogDataLoader(os.path.join(data_dir, 'test_data.csv'), dataset_train, device)

    model = RawGnn(device=device, dataset=dataset_train, embedding_size=Gs.embedding_size, gnn_layer_type=layer_type,
                   gnn_layer_count=layer_count, predictions=HemPredictionLayer, lambda_muq=Gs.lambda_muq_for_hem,
                   feature_interaction_order=order, phase2_attention=False).to(device)
    loss_function = nn.BCEWithLogitsLoss().to(device)
    from .optim import Adam
    grad_sync = None
    if world > 1:
        # one exchange step per training step: the batch rows' cotangents (no dense exchange: the default where the fused batch tail runs) | flat
        # all-reduce | per-bucket all-reduces overlapped with the backward | reduce-scatter + Adam on this rank's shard + all-gather.  Built BEFORE the optimizer: the
        # sharded exchange owns it.
        mode = args.grad_sync
        if mode in ('auto', '', None):
            batch_rows = Gs.batch_size * (1 + Gs.random_negative_sample_size + Gs.non_random_negative_sample_size)
            mode = ihg_dist.choose_gradient_sync(4 * sum(p.numel() for p in model.parameters()), world, model.supports_fused_loss(loss_function),
                                                 ihg_dist.cotangent_bytes_per_rank(batch_rows, model.compute_width * (layer_count + 1)))
            say(f'gradient exchange: {mode}')
        grad_sync = ihg_dist.make_gradient_sync(model, mode)
        grad_sync.broadcast_parameters(0)
    if grad_sync is not None and grad_sync.owns_optimizer:
        optimizer = grad_sync.optimizer(Gs.learning_rate, Gs.weight_decay)
    else:
        optimizer = Adam(model.parameters(), Gs.learning_rate, weight_decay=Gs.weight_decay)      # torch.optim.Adam's rule (Main.py:192), one launch

    epoch_start = 1
    if args.checkpoint:
        name = args.checkpoint
        if name == 'latest':
            found = sorted(fn for fn in os.listdir(result_dir)
                           if fn.startswith('checkpoint_') and os.path.isfile(os.path.join(result_dir, fn)))
            if not found:
                raise FileNotFoundError(f'no checkpoint_* file in {result_dir}')
            name = found[-1]
        state = torch.load(os.path.join(result_dir, name), map_location=device)
        model.load_state_dict(state['model'])
        optimizer.load_state_dict(state['optimizer'])
        epoch_start = int(state['epoch_count']) + 1
        say(f'resumed from {name} ({epoch_start - 1} epochs done)')

    if grad_sync is not None and args.checkpoint:
        grad_sync.broadcast_parameters(0)                    # (the loaded weights, identical on every rank anyway)

    store_from, store_every = (epoch_count, 1000000) if args.storecheckpoint else (None, None)
    if chief:
        say(f'\nmodel parameters ({len(list(model.parameters()))}):')
        print_network_parameters(model)
    pc = ProcessController(epoch_count, epoch_start, start_test, test_every, store_from, store_every)
    say(f'\nepochs {pc.EpochCount} | first test {start_test} | test every {test_every} | '
        f'first checkpoint {store_from} | checkpoint every {store_every}\n')

    history = MetricsCollection(Gs.use_valid_dataset)
    for _ in pc:
        avg_loss, train_seconds = train_and_get_avg_loss(model, optimizer, loss_function, dataset_train, dataloader_train,
                                                         pc, device, grad_sync=grad_sync, record_step=getattr(args, 'record_step', 'auto') if world == 1 else 'off')
        pc.AddTrainTime(train_seconds)
        if pc.ShouldStore():
            # every rank builds the state (a sharded optimizer gathers its Adam shards with a collective), the chief writes it
            state = ihg_dist.checkpoint_state(pc.CurrentEpoch, model, optimizer)
            if chief:
                fn = os.path.join(result_dir, time.strftime(f'checkpoint_%y%m%d-%H%M%S_epoch{pc.CurrentEpoch}', time.localtime()))
                say(f'\ncheckpoint -> {fn}')
                torch.save(state, fn)
            del state
        if pc.ShouldTest():
            say('\nevaluating on the TEST set ...')
            per_user, m_test, t_test = test_and_get_avg_metrics(model, dataset_train, dataloader_test, bool(Gs.long_tail_stat_fn))
            if chief and Gs.long_tail_stat_fn:
                with open(os.path.join(result_dir, Gs.long_tail_stat_fn), 'w', encoding='utf-8') as f:
                    for user, m in enumerate(per_user):
                        seen = int((dataset_train.pos_triples[:, 0] == user).sum())
                        tail = ',,,' if m is None else ',' + ','.join(m.to_string(no_title=True).split(' '))
                        f.write(f'{user},{seen}{tail}\n')
            if Gs.use_valid_dataset:
                say('evaluating on the VALIDATION set ...')
                _, m_valid, t_valid = test_and_get_avg_metrics(model, dataset_train, dataloader_valid)
                history.add(pc.CurrentEpoch, m_test, m_valid)
                pc.AddTestTime(t_test + t_valid)
            else:
                history.add(pc.CurrentEpoch, m_test)
                pc.AddTestTime(t_test)
            if chief and args.storemetrics:
                with open(fn_metrics, 'a', encoding='utf-8') as f:
                    f.write(f'Epoch {pc.CurrentEpoch} Avg loss {avg_loss:.4f}\n{m_test.to_string()}\n')

    if chief and len(list(history.iter_epoch_test())):
        by_ndcg = lambda m: m.NDCG_at10
        if Gs.use_valid_dataset:
            best_epoch, best_test, best_valid = history.get_valid_best(key=by_ndcg)
            say(f'best validation metrics at epoch \033[0;44m{best_epoch}\033[0m:')
            say(best_valid.to_string(highlight=True), put_time_in_single_line=True)
            say('test metrics at that epoch:')
        else:
            best_epoch, best_test = history.get_test_best(key=by_ndcg)
            best_valid = None
            say(f'best test metrics at epoch \033[0;44m{best_epoch}\033[0m:')
        say(best_test.to_string(highlight=True), put_time_in_single_line=True)
        if args.storemetrics:
            with open(fn_metrics, 'a', encoding='utf-8') as f:
                if best_valid is not None:
                    f.write(f'\n\nBest valid metrics at epoch {best_epoch}:\n{best_valid.to_string()}\nCorresponding test metrics:\n')
                else:
                    f.write('\nBest test metrics:\n')
                f.write(best_test.to_string() + '\n\n\nAll TEST metrics:\n' + f'Epoch {Metrics.title}\n')
                f.writelines(f'{e} {m.to_string(no_title=True)}\n' for e, m in history.iter_epoch_test())
                if Gs.use_valid_dataset:
                    f.write(f'\n\nAll VALID metrics:\nEpoch {Metrics.title}\n')
                    f.writelines(f'{e} {m.to_string(no_title=True)}\n' for e, _, m in history.iter_epoch_test_valid())
    # what the training loop decided about the step (TrainTestHelper.train_and_get_avg_loss, --record_step auto | on | off)
    history.training_step_recorded = bool(getattr(model, '_record_decision', False))
    history.eager_step_ms = getattr(model, '_auto_step_ms', None)
    if chief:
        IOHelper.EndLogging()
    return history


if __name__ == '__main__':
    main(sys.argv[1:])
