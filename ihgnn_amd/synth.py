"""Deterministic synthetic (user, query, item) search-log workloads.

The reference ships no data, so benchmarks, parity tests and the
golden-fixture generator all draw their inputs from here.  Two outputs are offered:

* :func:`draw` - the workload as numpy arrays (hyperedge member triples, query word bags,
  held-out logs); what ``bench.py`` feeds straight into ``GraphDataset.from_arrays``.
* :func:`write_files` - the same workload in the five on-disk files the reference's
  ``GraphDataset`` / ``TestSearchLogDataLoader`` parse (formats: ``Dataset.py:143-147``,
  ``Dataset.py:165-176``, ``Helpers/SearchLog.py:63-75`` of the reference).

Everything is a pure function of ``(shape, seed)`` through ``numpy.random.default_rng`` (PCG64).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

CSV_HEADER = 'user,query,search_time,items,pages,positions,interactions,times'


@dataclass
class Workload:
    """One synthetic personalised-search corpus.

    ``triples[e] = (user, query, item)`` are 0-based per-type ids of hyperedge ``e`` in file order.
    ``bag_words``/``bag_offsets`` are the flat 0-based word ids of every query and the start of
    each query's bag (``len(bag_offsets) == Q``).  ``eval_logs`` are held-out
    ``(user, query, [positive items])`` records for the valid/test files.
    """
    user_count: int
    query_count: int
    item_count: int
    vocab_size: int
    triples: np.ndarray            # [E, 3] int64
    bag_words: np.ndarray          # [sum words] int64, 0-based
    bag_offsets: np.ndarray        # [Q] int64
    valid_logs: List[Tuple[int, int, List[int]]] = field(default_factory=list)
    test_logs: List[Tuple[int, int, List[int]]] = field(default_factory=list)

    @property
    def node_count(self) -> int:
        return self.user_count + self.query_count + self.item_count

    @property
    def edge_count(self) -> int:
        return int(self.triples.shape[0])


def _zipf_ids(rng: np.random.Generator, count: int, size: int, exponent: float) -> np.ndarray:
    """``size`` draws from a truncated power law over ``count`` ids, hot ids scattered by a permutation."""
    ranks = np.arange(1, count + 1, dtype=np.float64)
    weights = ranks ** (-exponent)
    cdf = np.cumsum(weights)
    cdf /= cdf[-1]
    picks = np.searchsorted(cdf, rng.random(size), side='left')
    np.minimum(picks, count - 1, out=picks)
    scatter = rng.permutation(count)
    return scatter[picks].astype(np.int64)


def draw(user_count: int, query_count: int, item_count: int, vocab_size: int, edge_count: int,
         seed: int = 0, distribution: str = 'uniform', exponent: float = 1.2,
         max_words: int = 5, eval_logs: int = 0) -> Workload:
    """Draw a workload.  ``distribution`` is ``'uniform'`` (config C1) or ``'powerlaw'`` (config C5)."""
    rng = np.random.default_rng(seed)
    if distribution == 'uniform':
        u = rng.integers(0, user_count, edge_count, dtype=np.int64)
        q = rng.integers(0, query_count, edge_count, dtype=np.int64)
        i = rng.integers(0, item_count, edge_count, dtype=np.int64)
    elif distribution == 'powerlaw':
        u = _zipf_ids(rng, user_count, edge_count, exponent)
        q = _zipf_ids(rng, query_count, edge_count, exponent)
        i = _zipf_ids(rng, item_count, edge_count, exponent)
    else:
        raise ValueError(f'unknown distribution {distribution!r}')
    triples = np.stack([u, q, i], axis=1)

    lengths = rng.integers(1, max_words + 1, query_count, dtype=np.int64)
    offsets = np.zeros(query_count, dtype=np.int64)
    np.cumsum(lengths[:-1], out=offsets[1:])
    words = rng.integers(0, vocab_size, int(lengths.sum()), dtype=np.int64)

    def held_out(n: int) -> List[Tuple[int, int, List[int]]]:
        logs = []
        for _ in range(n):
            k = int(rng.integers(1, 4))
            items = sorted(set(int(x) for x in rng.integers(0, item_count, k)))
            logs.append((int(rng.integers(0, user_count)), int(rng.integers(0, query_count)), items))
        return logs

    return Workload(user_count, query_count, item_count, vocab_size, triples, words, offsets,
                    held_out(eval_logs), held_out(eval_logs))


def _log_row(user: int, query: int, items: Sequence[int], flags: Sequence[int], stamp: int) -> str:
    n = len(items)
    cols = [str(user), str(query), str(stamp),
            ' '.join(map(str, items)),
            ' '.join(['1'] * n),
            ' '.join(str(p + 1) for p in range(n)),
            ' '.join(map(str, flags)),
            ' '.join(str(stamp) if f > 0 else 'NA' for f in flags)]
    return ','.join(cols)


def write_files(w: Workload, directory: str,
                train_rows: Optional[List[Tuple[int, int, List[int], List[int]]]] = None) -> dict:
    """Write ``w`` as graph_info.txt / queries_multihot.txt / {train,valid,test}_data.csv.

    By default every hyperedge becomes its own one-item log row (file order == hyperedge order).
    ``train_rows`` overrides the training file with explicit ``(user, query, items, flags)`` rows,
    which is how the tiny hand-made fixture gets multi-item logs, negatives and duplicates.
    Returns the five paths keyed like the reference's constructor arguments.
    """
    os.makedirs(directory, exist_ok=True)
    paths = {k: os.path.join(directory, v) for k, v in dict(
        fn_graph_info='graph_info.txt', fn_queries_multihot='queries_multihot.txt',
        fn_train_data='train_data.csv', fn_valid_data='valid_data.csv', fn_test_data='test_data.csv').items()}

    with open(paths['fn_graph_info'], 'w', encoding='utf-8') as f:
        f.write(f'{w.user_count} {w.query_count} {w.item_count} {w.vocab_size}\n')

    ends = np.append(w.bag_offsets[1:], len(w.bag_words))
    with open(paths['fn_queries_multihot'], 'w', encoding='utf-8') as f:
        for a, b in zip(w.bag_offsets, ends):
            f.write(' '.join(map(str, w.bag_words[a:b].tolist())) + '\n')

    stamp0 = 1_400_000_000
    with open(paths['fn_train_data'], 'w', encoding='utf-8') as f:
        f.write(CSV_HEADER + '\n')
        if train_rows is None:
            for e, (u, q, i) in enumerate(w.triples.tolist()):
                f.write(_log_row(u, q, [i], [1], stamp0 + e) + '\n')
        else:
            for e, (u, q, items, flags) in enumerate(train_rows):
                f.write(_log_row(u, q, items, flags, stamp0 + e) + '\n')

    for key, logs in (('fn_valid_data', w.valid_logs), ('fn_test_data', w.test_logs)):
        with open(paths[key], 'w', encoding='utf-8') as f:
            f.write(CSV_HEADER + '\n')
            for e, (u, q, items) in enumerate(logs):
                f.write(_log_row(u, q, items, [1] * len(items), stamp0 + 10_000_000 + e) + '\n')
    return paths


# Named shapes of BASELINE.json's configs.  C2-C4 name real Amazon / CIKM corpora that are not in
# the image; their entries are size-matched synthetic stand-ins and are labelled as such wherever
# they are reported (bench.py ``config.workload``).
CONFIGS = {
    'C1': dict(user_count=1000, query_count=500, item_count=1000, vocab_size=300, edge_count=20_000,
               seed=0, distribution='uniform', dim=64, layers=1),
    # Amazon Electronics 5-core has ~192k users / ~63k items / ~1.69M reviews; one query per
    # category path (~1k).  Train split ~ 80 %.
    'C2': dict(user_count=192_403, query_count=989, item_count=63_001, vocab_size=1_200, edge_count=1_350_000,
               seed=2, distribution='powerlaw', exponent=0.8, dim=64, layers=2),
    'C3': dict(user_count=230_000, query_count=26_000, item_count=120_000, vocab_size=30_000, edge_count=2_200_000,
               seed=3, distribution='powerlaw', exponent=0.8, dim=128, layers=3),
    # "Amazon full catalog" (BASELINE configs[3]): the union of the five Amazon 5-core corpora the reference's driver lists
    # (Main.py:35-39: OfficeProducts, CellPhones, KindleStore, VideoGames, Electronics).  Their published 5-core counts add up
    # to 317,713 users / 148,456 items / 3,151,284 reviews; Step1-Amazon.py:126-134 writes one log per (review, category-path
    # query) - about 1.3 per review - and the train split keeps ~80 %: E = 3.3 M.  Queries = distinct category paths (~5 k).
    # One replica of this graph per GPU (data-parallel batches); the 8-GPU run is `bench.py --config C4 --gpus 8`.
    'C4': dict(user_count=317_713, query_count=5_000, item_count=148_456, vocab_size=6_000, edge_count=3_300_000,
               seed=4, distribution='powerlaw', exponent=0.8, dim=128, layers=3),
    'C5': dict(user_count=4_000_000, query_count=1_000_000, item_count=5_000_000, vocab_size=200_000,
               edge_count=50_000_000, seed=5, distribution='powerlaw', exponent=1.2, dim=256, layers=2),
}


def draw_config(name: str, scale: float = 1.0, eval_logs: int = 0) -> Workload:
    """Draw one of :data:`CONFIGS`; ``scale`` < 1 shrinks every count (CPU-baseline subsamples)."""
    c = dict(CONFIGS[name])
    c.pop('dim'), c.pop('layers')
    for k in ('user_count', 'query_count', 'item_count', 'vocab_size', 'edge_count'):
        c[k] = max(4, int(round(c[k] * scale)))
    return draw(eval_logs=eval_logs, **c)
