"""Command line of the driver: the reference's flags, aliases and defaults (``Helpers/ArgsParser.py:49-96``)."""
import argparse
from typing import Optional, Sequence

from .GlobalSettings import Gsv

# (dest, flags, kind, default, help); kind is a type or 'flag'
_FLAGS = (
    ('checkpoint', ('--checkpoint', '--cp'), str, '', 'checkpoint file inside the result dir, or "latest"; empty = fresh start'),
    ('storecheckpoint', ('--storecheckpoint', '--scp', '-c'), 'flag', False, 'write a checkpoint when the schedule says so'),
    ('storemetrics', ('--storemetrics', '--sm', '-m'), 'flag', False, 'append test metrics to <time>_metrics.txt'),
    ('epoch_count', ('--epoch_count', '--ec'), int, 0, 'epochs to run (0 = driver default 110)'),
    ('epoch_start_test', ('--epoch_start_test', '--est'), int, 0, 'first epoch to test at (0 = driver default 10)'),
    ('epoch_test_frequency', ('--epoch_test_frequency', '--etf'), int, 0, 'test every this many epochs'),
    ('dataset', ('--dataset', '--ds'), str, '', 'dataset sub-directory under the data root'),
    ('model', ('--model',), str, '', 'model type (RawGnn)'),
    ('gnn', ('--gnn',), str, '', 'GNN layer type: IHGNN | HGCN (GCN / GAT are not part of this build)'),
    ('gnns', ('--gnns',), int, 0, 'number of GNN layers (0 = driver default 2)'),
    ('feature_order', ('--feature_order', '--fo'), int, 0, 'interaction order 1 | 2 | 3 (0 = driver default 3)'),
    ('completeness', ('--completeness',), str, Gsv.graph_uqi, 'pairwise-graph completeness (GCN only)'),
    ('longtail', ('--longtail',), str, '', 'per-user long-tail statistics file name'),
    ('device', ('--device', '-d'), str, '', 'GPU ordinal ("0" = cuda:0)'),
    ('embedding_size', ('--embedding_size', '--emb'), int, 0, 'embedding width (0 = Gs.embedding_size)'),
    # not in the reference: batches (positive permutation + negative sampling) produced on the GPU instead of by DataLoader + random.sample
    ('device_sampling', ('--device_sampling',), 'flag', False, 'draw training batches on the device (same distribution, different random stream)'),
    ('grad_sync', ('--grad_sync',), str, 'auto', 'gradient exchange under torchrun: auto | cotangent (batch-row cotangents: no dense exchange) | flat | bucketed | sharded (ihgnn_amd.distributed)'),
    ('seed', ('--seed',), int, -1, 'seed torch / random / numpy before the model is built (the reference seeds nothing, Main.py: -1 leaves the generators alone)'),
    ('record_step', ('--record_step',), 'optional', 'auto', 'replay the training step as one recorded hipGraph (single process): auto (default: when an eager step measures launch-bound, '
                                                             '< 1.5 ms) | on (also a bare --record_step) | off'),
)


class ConsoleArgs:
    def __init__(self, ns: argparse.Namespace):
        for dest, *_ in _FLAGS:
            setattr(self, dest, getattr(ns, dest))
        self.long_tail_filename = ns.longtail


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description='IHGNN on MI355X')
    for dest, flags, kind, default, text in _FLAGS:
        if kind == 'flag':
            parser.add_argument(*flags, dest=dest, action='store_true', default=default, help=text)
        elif kind == 'optional':
            parser.add_argument(*flags, dest=dest, nargs='?', const='on', default=default, choices=('auto', 'on', 'off'), help=text)
        else:
            parser.add_argument(*flags, dest=dest, type=kind, default=default, help=text)
    return parser


def parse_args(argv: Optional[Sequence[str]] = None) -> ConsoleArgs:
    return ConsoleArgs(build_parser().parse_args(argv))
