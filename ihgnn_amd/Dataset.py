"""Training / evaluation data feeding the hypergraph path (reference ``Dataset.py``).

``GraphDataset`` keeps the reference's constructor, attributes, ``__getitem__`` / ``collate_fn`` contract and
lazy graph properties (``Dataset.py:11-293``) and adds ``from_arrays`` for in-memory synthetic corpora.
Positive interactions are held as one ``[E,3]`` int64 array (file order = hyperedge order); the reference's
list-of-namedtuples view is built only if somebody asks for ``pos_interactions``.
"""
from __future__ import annotations

import os
import random
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import Dataset

from .Helpers.Graph import Pps2DGraph, PpsGraph, PpsHyperGraph, PpsLogHyperGraph
from .Helpers.IOHelper import IOHelper
from .Helpers.SearchLog import PosInteraction, SearchLog
from .Helpers.SearchLogCollection import SearchLogCollection

Sample = Tuple[Tuple[int, int, int, int], List[int]]


def parse_search_logs(filename: str, with_rows: bool = False):
    """Native two-pass parse of a search-log CSV -> (row count, positive ``[P,3]``, negative ``[M,3]`` (user, query, item));
    ``with_rows`` appends the 0-based row of every positive (``[P]`` int64: positives of one search log share it)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    path = os.fsencode(filename)
    n_logs, n_pos, n_neg = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.ihg_parse_search_logs(path, ctypes.byref(n_logs), ctypes.byref(n_pos), ctypes.byref(n_neg), None, 0, None, 0, None),
               'ihg_parse_search_logs')
    pos = np.empty((max(n_pos.value, 1), 3), np.int64)
    neg = np.empty((max(n_neg.value, 1), 3), np.int64)
    rows = np.empty(max(n_pos.value, 1), np.int64)
    as_ptr = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    _lib.check(lib.ihg_parse_search_logs(path, ctypes.byref(n_logs), ctypes.byref(n_pos), ctypes.byref(n_neg),
                                         as_ptr(pos), n_pos.value, as_ptr(neg), n_neg.value, as_ptr(rows)), 'ihg_parse_search_logs')
    out = (int(n_logs.value), pos[:n_pos.value], neg[:n_neg.value])
    return out + (rows[:n_pos.value],) if with_rows else out


def read_graph_info(filename: str) -> List[int]:
    """``graph_info.txt`` -> ``[users, queries, items, vocabulary]`` (native reader; ``Dataset.py:143-147`` of the reference)."""
    import ctypes
    from . import _lib
    counts = (ctypes.c_int64 * 4)()
    _lib.check(_lib.load().ihg_read_graph_info(os.fsencode(filename), counts), 'ihg_read_graph_info')
    return [int(c) for c in counts]


def read_query_bags(filename: str) -> Tuple[np.ndarray, np.ndarray]:
    """``queries_multihot.txt`` -> (flat 0-based word ids, start offset of every query) (native two-pass reader;
    ``Dataset.py:165-176`` of the reference)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    path = os.fsencode(filename)
    n_q, n_w = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.ihg_read_query_bags(path, ctypes.byref(n_q), ctypes.byref(n_w), None, 0, None, 0), 'ihg_read_query_bags')
    offsets = np.empty(max(n_q.value, 1), np.int64)
    words = np.empty(max(n_w.value, 1), np.int64)
    as_ptr = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    _lib.check(lib.ihg_read_query_bags(path, ctypes.byref(n_q), ctypes.byref(n_w), as_ptr(offsets), n_q.value, as_ptr(words), n_w.value),
               'ihg_read_query_bags')
    return words[:n_w.value], offsets[:n_q.value]


class GraphDataset(Dataset):
    device: torch.device = None        # class attribute, as in the reference (Dataset.py:135): collate_fn is static

    def __init__(self, fn_graph_info: str, fn_queries_multihot: str, fn_train_data: str, graph_type: type,
                 random_negative_sample_size: int, non_random_negative_sample_size: int, device: torch.device):
        super().__init__()
        counts = read_graph_info(fn_graph_info)                      # the three input files go through the native readers
        bag_words, bag_offsets = read_query_bags(fn_queries_multihot)
        n_logs, triples, negatives, rows = parse_search_logs(fn_train_data, with_rows=True)
        self._setup(counts, bag_words, bag_offsets, triples, graph_type,
                    random_negative_sample_size, non_random_negative_sample_size, device)
        self.pos_log = rows
        self._fn_train_data = fn_train_data
        self._search_logs = None
        self._neg_triples = negatives
        self._neg_interactions = None
        self._neg_of = None

        IOHelper.LogPrint(f'training set ready: {fn_train_data}')
        IOHelper.LogPrint(f'users {self.user_count} | queries {self.query_count} | items {self.item_count} | '
                          f'vocabulary {self.vocab_size} | logs {n_logs} | hyperedges {len(self)} | {graph_type.__name__}')
        if len(self):
            IOHelper.LogPrint(f'{len(negatives) / len(self):.4f} logged negatives per positive')

    @classmethod
    def from_arrays(cls, user_count: int, query_count: int, item_count: int, vocab_size: int,
                    bag_words: np.ndarray, bag_offsets: np.ndarray, triples: np.ndarray,
                    graph_type: type = PpsHyperGraph, random_negative_sample_size: int = 10,
                    non_random_negative_sample_size: int = 0, device: torch.device = torch.device('cuda:0'),
                    pos_log: Optional[np.ndarray] = None) -> 'GraphDataset':
        """In-memory construction (synthetic corpora): ``bag_words`` are 0-based word ids, ``triples`` 0-based per type;
        ``pos_log[e]`` = the search log hyperedge ``e`` came from (default: one log per positive)."""
        self = cls.__new__(cls)
        Dataset.__init__(self)
        self._setup([user_count, query_count, item_count, vocab_size], np.asarray(bag_words, np.int64),
                    np.asarray(bag_offsets, np.int64), np.asarray(triples, np.int64).reshape(-1, 3), graph_type,
                    random_negative_sample_size, non_random_negative_sample_size, device)
        if pos_log is not None:
            self.pos_log = np.ascontiguousarray(pos_log, dtype=np.int64)
        self._fn_train_data = None
        self._search_logs = None
        self._neg_triples = np.zeros((0, 3), np.int64)
        self._neg_interactions = None
        self._neg_of = None
        return self

    def _setup(self, counts: Sequence[int], bag_words: np.ndarray, bag_offsets: np.ndarray, triples: np.ndarray,
               graph_type: type, rand_neg: int, nonrand_neg: int, device: torch.device) -> None:
        if graph_type not in (Pps2DGraph, PpsHyperGraph, PpsLogHyperGraph):
            raise AssertionError(f'unsupported graph type: {graph_type}')
        GraphDataset.device = device
        self.graph_type = graph_type
        self.rand_neg_sample_size = rand_neg
        self.nonrand_neg_sample_size = nonrand_neg
        self.neg_sample_size = rand_neg + nonrand_neg

        self.user_count, self.query_count, self.item_count, self.vocab_size = (int(c) for c in counts)
        self.node_count = self.user_count + self.query_count + self.item_count
        self.query_start_index_in_graph = self.user_count
        self.item_start_index_in_graph = self.user_count + self.query_count
        if bag_offsets.shape[0] != self.query_count:
            raise ValueError(f'{bag_offsets.shape[0]} query bags for {self.query_count} queries')

        # ids are 0-based on disk; row 0 of every embedding table is padding, so stored id = index + 1
        self.users_onehot = torch.arange(1, 1 + self.user_count, device=device)
        self.items_onehot = torch.arange(1, 1 + self.item_count, device=device)
        self.vocabulary_onehot = torch.arange(1, 1 + self.vocab_size, device=device)
        self.bag_words_host = bag_words + 1
        self.bag_offsets_host = bag_offsets
        self.queries_for_embeddingbag = torch.from_numpy(self.bag_words_host).to(device)
        self.queries_offset_for_embeddingbag = torch.from_numpy(bag_offsets).to(device)

        self.pos_triples = np.ascontiguousarray(triples)
        self._pos_interactions: Optional[List[PosInteraction]] = None
        self._hgraph: Optional[PpsHyperGraph] = None
        self._hloggraph: Optional[PpsLogHyperGraph] = None
        self.pos_log = np.arange(self.pos_triples.shape[0], dtype=np.int64)      # search log (file row) of every positive; from_arrays: one each
        self._graph2d = None
        self._bag_layout = None
        self._queries_multihot = None

    # -- lazily built views -----------------------------------------------------------------------
    @property
    def search_logs(self) -> Optional[SearchLogCollection]:
        """The parsed rows as the reference keeps them (``Dataset.py:192``); only materialised if somebody asks."""
        if self._search_logs is None and self._fn_train_data is not None:
            self._search_logs = SearchLogCollection.read(self._fn_train_data)
        return self._search_logs

    @property
    def neg_interactions(self) -> List[Tuple[int, int, int]]:
        if self._neg_interactions is None:
            self._neg_interactions = [tuple(r) for r in self._neg_triples.tolist()]
        return self._neg_interactions

    @property
    def neg_items_for_user_query_pair(self) -> Dict[Tuple[int, int], List[int]]:
        """(user, query) -> logged negative items in file order; every logged pair has an entry (``Dataset.py:201-210``)."""
        if self._neg_of is None:
            table: Dict[Tuple[int, int], List[int]] = {}
            if self.search_logs is not None:
                for log in self.search_logs:
                    bucket = table.setdefault((log.user, log.query), [])
                    bucket.extend(item for item, flag in zip(log.items, log.interactions) if flag <= 0)
            self._neg_of = table
        return self._neg_of

    @property
    def pos_interactions(self) -> List[PosInteraction]:
        if self._pos_interactions is None:
            if self.search_logs is not None:
                self._pos_interactions = [p for log in self.search_logs for p in PosInteraction.from_search_log(log, True)]
            else:
                self._pos_interactions = [PosInteraction(u, q, '', i, 1, 1, 1, '') for u, q, i in self.pos_triples.tolist()]
        return self._pos_interactions

    @property
    def queries_multihot(self) -> Tensor:
        """Sparse ``[Q x V]`` row-normalised bag-of-words matrix (``Dataset.py:178-183``); unused by the models."""
        if self._queries_multihot is None:
            ends = np.append(self.bag_offsets_host[1:], self.bag_words_host.shape[0])
            lens = ends - self.bag_offsets_host
            rows = np.repeat(np.arange(self.query_count), lens)
            vals = np.repeat(1.0 / np.maximum(lens, 1), lens).astype(np.float32)
            self._queries_multihot = torch.sparse_coo_tensor(
                np.stack([rows, self.bag_words_host - 1]), vals, (self.query_count, self.vocab_size)).coalesce().to(GraphDataset.device)
        return self._queries_multihot

    @property
    def bag_layout(self):
        """Device layout of the query bags for the HIP embedding-bag kernels."""
        if self._bag_layout is None:
            from .ops import BagLayout
            self._bag_layout = BagLayout(self.bag_words_host, self.bag_offsets_host, self.vocab_size + 1, GraphDataset.device)
        return self._bag_layout

    @property
    def hypergraph(self) -> PpsHyperGraph:
        if self._hgraph is None:
            self._hgraph = PpsHyperGraph.from_triples(self.pos_triples, self.node_count, self.user_count,
                                                      self.query_count, GraphDataset.device)
        return self._hgraph

    @property
    def graph2d(self) -> Pps2DGraph:
        if self._graph2d is None:
            self._graph2d = Pps2DGraph.from_triples(self.pos_triples, self.node_count, self.user_count, self.query_count,
                                                    False, GraphDataset.device)
        return self._graph2d

    @property
    def hypergraph_log(self) -> PpsLogHyperGraph:
        """One variable-arity hyperedge per search log with a positive (``Dataset.py:98-103``, ``Graph.py:138-189``)."""
        if self._hloggraph is None:
            self._hloggraph = PpsLogHyperGraph.from_positives(self.pos_triples, self.pos_log, self.node_count, self.user_count,
                                                              self.query_count, GraphDataset.device)
        return self._hloggraph

    @property
    def graph(self) -> PpsGraph:
        if self.graph_type == Pps2DGraph:
            return self.graph2d
        return self.hypergraph_log if self.graph_type == PpsLogHyperGraph else self.hypergraph

    # -- sampling ------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return int(self.pos_triples.shape[0])

    def __getitem__(self, index: int) -> Sample:
        """One positive and its sampled negative items (``Dataset.py:107-119``): ``random.sample`` draws distinct
        items per call and may hit the positive; logged negatives of the (user, query) pair come first if requested."""
        u, q, i = (int(x) for x in self.pos_triples[index])
        want_logged = self.nonrand_neg_sample_size
        if want_logged == 0:
            return (u, q, i, 1), random.sample(range(self.item_count), self.rand_neg_sample_size)
        logged = self.neg_items_for_user_query_pair.get((u, q), [])
        if len(logged) < want_logged:
            return (u, q, i, 1), random.sample(range(self.item_count), self.neg_sample_size - len(logged)) + logged
        chosen = random.sample(logged, want_logged)
        return (u, q, i, 1), chosen + random.sample(range(self.item_count), self.rand_neg_sample_size)

    @staticmethod
    def collate_fn(data: List[Sample]) -> Tuple[Tensor, ...]:
        """-> (users, queries, items, flags) of the positives then of the negatives, on ``GraphDataset.device``.

        Everything is packed into one host array and crosses PCIe once (the reference builds eight tensors from
        Python lists, one copy each, ``Dataset.py:282-291``)."""
        n_pos = len(data)
        n_neg = sum(len(negs) for _, negs in data)
        pack = np.zeros((4, n_pos + n_neg), dtype=np.int64)
        cursor = n_pos
        for k, ((u, q, item, flag), negs) in enumerate(data):
            pack[:, k] = (u, q, item, flag)
            stop = cursor + len(negs)
            pack[0, cursor:stop] = u
            pack[1, cursor:stop] = q
            pack[2, cursor:stop] = negs
            cursor = stop
        dev = _to_device_async(pack, GraphDataset.device)
        pos, neg = dev[:, :n_pos], dev[:, n_pos:]
        return pos[0], pos[1], pos[2], pos[3], neg[0], neg[1], neg[2], neg[3]

    def device_batches(self, batch_size: int, epoch: int = 0, seed: int = 0, rank: int = 0, world_size: int = 1):
        """One epoch of training batches produced ON THE DEVICE: the positives are a device-side permutation of the hyperedges
        (this rank's share of it when ``world_size`` > 1), the ``rand_neg_sample_size`` negatives of every positive come from
        ``ihg_sample_negatives`` (distinct within a sample, uniform over the catalogue, may hit the positive - the semantics of
        ``random.sample(range(I), k)``, ``Dataset.py:107-109``).  Yields the 8-tuple of ``collate_fn`` (positives then negatives;
        users, queries, items, flags) with no host -> device copy and no host-side sampling per step.  Logged (non-random)
        negatives are a host-side table and are not drawn here."""
        import ctypes
        from . import _lib
        if self.nonrand_neg_sample_size:
            raise ValueError('device_batches draws random negatives only (non_random_negative_sample_size must be 0)')
        lib = _lib.load()
        dev = GraphDataset.device
        if getattr(self, '_pos_device', None) is None:
            self._pos_device = torch.from_numpy(self.pos_triples).to(dev)
        k = self.rand_neg_sample_size
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed * 1_000_003 + epoch)
        order = torch.randperm(len(self), device=dev, generator=gen)[rank::world_size]
        n_batches = -(-(-(-len(self) // world_size)) // batch_size)
        mine = int(order.shape[0])
        if mine < n_batches:
            raise ValueError(f'rank {rank} holds {mine} positives for {n_batches} batches: batch_size {batch_size} is too small for {world_size} ranks '
                             '(an empty batch on one rank would leave the others waiting in the gradient exchange)')
        base, extra = divmod(mine, n_batches)
        lo = 0
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for b in range(n_batches):
            if world_size == 1:                                          # DataLoader's batching: full batches and a shorter last one
                hi = min(lo + batch_size, mine)
            else:                                                        # equal batch counts on every rank, sizes within one row of each other
                hi = lo + base + (1 if b < extra else 0)
            pos = self._pos_device[order[lo:hi]]
            lo = hi
            n = int(pos.shape[0])
            neg_items = torch.empty(n, k, dtype=torch.int64, device=dev)
            _lib.check(lib.ihg_sample_negatives(seed * 7919 + rank, (epoch << 32) + b, n, self.item_count, k, ctypes.c_void_p(neg_items.data_ptr()), stream),
                       'ihg_sample_negatives')
            ones = torch.ones(n, dtype=torch.int64, device=dev)
            yield (pos[:, 0], pos[:, 1], pos[:, 2], ones, pos[:, 0].repeat_interleave(k), pos[:, 1].repeat_interleave(k),
                   neg_items.reshape(-1), torch.zeros(n * k, dtype=torch.int64, device=dev))

    def sample_batches(self, batch_size: int, steps: int, seed: int = 0) -> Iterator[Tuple[Tensor, Tensor, Tensor, Tensor]]:
        """Vectorised batch source for benchmarks: ``steps`` batches of ``batch_size`` positives (uniform with
        replacement) each followed by ``rand_neg_sample_size`` uniform negatives; yields device tensors
        ``(users, queries, items, labels)`` of ``batch_size * (1 + negatives)`` rows."""
        rng = np.random.default_rng(seed)
        k = self.rand_neg_sample_size
        for _ in range(steps):
            pick = rng.integers(0, len(self), batch_size)
            pos = self.pos_triples[pick]
            neg_items = rng.integers(0, self.item_count, batch_size * k)
            users = np.concatenate([pos[:, 0], np.repeat(pos[:, 0], k)])
            queries = np.concatenate([pos[:, 1], np.repeat(pos[:, 1], k)])
            items = np.concatenate([pos[:, 2], neg_items])
            labels = np.concatenate([np.ones(batch_size, np.float32), np.zeros(batch_size * k, np.float32)])
            dev = GraphDataset.device
            yield (torch.from_numpy(users).to(dev), torch.from_numpy(queries).to(dev),
                   torch.from_numpy(items).to(dev), torch.from_numpy(labels).to(dev))


class DeviceBatchLoader:
    """Training-batch source that never leaves the GPU (``GraphDataset.device_batches``): drop-in for the ``DataLoader`` of the
    training loop (iterating yields the ``collate_fn`` 8-tuple; ``batch_sampler.set_epoch`` selects the epoch's permutation)."""

    def __init__(self, dataset: GraphDataset, batch_size: int, rank: int = 0, world_size: int = 1, seed: int = 0):
        self.dataset, self.batch_size, self.rank, self.world_size, self.seed = dataset, batch_size, rank, world_size, seed
        self.epoch = 0
        self.batch_sampler = self

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return -(-(-(-len(self.dataset) // self.world_size)) // self.batch_size)

    def __iter__(self):
        return self.dataset.device_batches(self.batch_size, self.epoch, self.seed, self.rank, self.world_size)


class _PinnedRing:
    """A few page-locked staging buffers reused round-robin: a pageable ``.to(device)`` blocks the host until every kernel
    queued before it has run, which serialises the training loop with the GPU once per step; a pinned, non-blocking copy does
    not.  A slot is reused only after the copy that last read it has executed (event per slot)."""

    SLOTS = 4

    def __init__(self):
        self.buffers = [None] * self.SLOTS
        self.events = [None] * self.SLOTS
        self.cursor = 0

    def stage(self, array: np.ndarray) -> Tensor:
        k = self.cursor
        self.cursor = (k + 1) % self.SLOTS
        if self.events[k] is not None:
            self.events[k].synchronize()
        n = array.size
        buf = self.buffers[k]
        if buf is None or buf.numel() < n or buf.dtype != torch.from_numpy(array).dtype:
            buf = self.buffers[k] = torch.empty(max(n, 4096), dtype=torch.from_numpy(array).dtype).pin_memory()
        view = buf[:n].view(array.shape)
        view.copy_(torch.from_numpy(array))
        return view

    def mark(self, k_stream_event) -> None:
        self.events[(self.cursor - 1) % self.SLOTS] = k_stream_event


_RING = None


def _to_device_async(array: np.ndarray, device: torch.device) -> Tensor:
    """Host array -> device tensor without stalling the host behind the GPU's queue (see ``_PinnedRing``)."""
    global _RING
    if device.type != 'cuda':
        return torch.from_numpy(array).to(device)
    if _RING is None:
        _RING = _PinnedRing()
    staged = _RING.stage(np.ascontiguousarray(array))
    out = staged.to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _RING.mark(ev)
    return out


class TestSearchLogDataLoader:
    """Evaluation logs: every search with >= 1 positive, as (user, query, de-duplicated positive items).

    Iteration yields the reference's 5-tuple (``Dataset.py:324-329``).  ``users`` / ``queries`` are length-``I``
    index vectors like the reference's ``u * ones(I)``, but as stride-0 expansions of a single element, so the
    model can tell "one (user, query) against every item" apart and score it as a matrix-vector product.
    """
    __test__ = False          # not a pytest class

    def __init__(self, fn_search_log: str, dataset_train: GraphDataset, device: torch.device):
        self.logs: List[Tuple[int, int, List[int], Optional[List[int]], bool]] = []
        rows = 0
        with open(fn_search_log, 'r', encoding='utf-8') as f:
            next(f, None)
            for line in f:
                if not line.strip():
                    continue
                rows += 1
                log = SearchLog.parse(line)
                if sum(log.interactions) > 0:
                    self.logs.append((log.user, log.query, log.get_interacted_items()[0], None, True))
        self.item_count = dataset_train.item_count
        self.device = device
        IOHelper.LogPrint(f'evaluation set ready: {fn_search_log} ({rows} rows, {len(self.logs)} usable logs)')

    def __len__(self) -> int:
        return len(self.logs)

    def __iter__(self):
        if not self.logs:
            return
        uq = torch.tensor([(u, q) for u, q, *_ in self.logs], dtype=torch.long, device=self.device)
        for k, (_, _, items, flags, all_1) in enumerate(self.logs):
            yield uq[k, 0:1].expand(self.item_count), uq[k, 1:2].expand(self.item_count), items, flags, all_1
