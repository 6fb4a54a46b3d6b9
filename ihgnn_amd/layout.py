"""HBM-resident incidence layouts consumed by the HIP kernels.

* :class:`Csr` - a row -> id-list structure (``ptr[n_rows+1]``, ``ids[nnz]``, int32) plus the split-row
  plan for skewed rows (rows longer than ``heavy_threshold`` are cut into ``heavy_chunk``-sized segments
  summed by separate lane groups; see ``ihg_node_segment_sum_heavy``).
* :class:`IncidenceLayout` - the (user, query, item) hypergraph: ``i3[E,3]`` edge-major member ids (fixed
  arity, so no edge-side row pointer) and the node-major CSR, built by the native ``ihg_build_csr``.

Replaces the coalesced COO ``Adjacency`` + int64 ``I3`` of ``Helpers/Graph.py:94-134`` (reference) as the
thing kernels read; 12 B of int32 ids per hyperedge on each side instead of 48 B of int64 COO.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib

import os

# Split-row plan: rows longer than a THRESHOLD are cut into segments of CHUNK ids (each summed by a lane group of its own; the partial sums added in a fixed tree).
# IHG_HEAVY_THRESHOLD / IHG_HEAVY_CHUNK fix them; unset, the plan follows the list's size: 256 / 128 below LARGE_LIST entries, 512 / 512 from there on - on the BASELINE-size
# lists fewer, longer segments are worth a little everywhere (round 6, profiles/r6/09_ab_split_row_plan.txt: pair sums 772 -> 745 us at C3, 206 -> 187 at C2, 1,060 -> 1,003 at C4;
# step C3 - 0.6 %, C2 - 1.0 %, C5 - 0.4 %, HGCN layers at C3 5.43 -> 5.30 ms), while the small graphs of tests and examples keep split rows on their path.
_env_threshold, _env_chunk = os.environ.get('IHG_HEAVY_THRESHOLD'), os.environ.get('IHG_HEAVY_CHUNK')
HEAVY_THRESHOLD = int(_env_threshold) if _env_threshold is not None else None      # None: by the list's size (split_row_plan_for)
HEAVY_CHUNK = int(_env_chunk) if _env_chunk is not None else None
LARGE_LIST = 1 << 20


def split_row_plan_for(nnz: int, heavy_threshold: Optional[int] = None, heavy_chunk: Optional[int] = None) -> Tuple[int, int]:
    """``(threshold, chunk)`` of a list of ``nnz`` entries: the caller's, else the environment's, else by size (256 / 128 below ``LARGE_LIST`` entries, 512 / 512 above)."""
    threshold = heavy_threshold if heavy_threshold is not None else HEAVY_THRESHOLD
    if threshold is None:
        threshold = 512 if nnz >= LARGE_LIST else 256
    chunk = heavy_chunk if heavy_chunk is not None else HEAVY_CHUNK
    if chunk is None:
        chunk = 512 if threshold >= 512 else 128
    return int(threshold), int(chunk)


HEAVY_MAX_SEGMENTS = int(os.environ.get('IHG_HEAVY_MAX_SEGMENTS', 4096))   # ... but never more than this many per row (longer segments instead; C5 step, ms: uncapped 683, 4096: 661, 1024: 667, 256: 691)


def _as_ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


class Csr:
    """Device-resident CSR with an optional split-row plan."""

    def __init__(self, ptr_host: np.ndarray, ids_host: np.ndarray, device: torch.device,
                 heavy_threshold: Optional[int] = None, heavy_chunk: Optional[int] = None):
        ptr_host = np.ascontiguousarray(ptr_host, dtype=np.int32)
        ids_host = np.ascontiguousarray(ids_host, dtype=np.int32)
        heavy_threshold, heavy_chunk = split_row_plan_for(int(ids_host.shape[0]), heavy_threshold, heavy_chunk)
        self.n_rows = int(ptr_host.shape[0] - 1)
        self.nnz = int(ids_host.shape[0])
        self.device = device
        self.ptr_host, self.ids_host = ptr_host, ids_host
        self.ptr = torch.from_numpy(ptr_host).to(device)
        self.ids = torch.from_numpy(ids_host).to(device)
        self.heavy_threshold = int(heavy_threshold)
        self._plan_heavy(int(heavy_chunk))
        self._plan_order()
        self._partials: Dict[Tuple[int, int], torch.Tensor] = {}

    def _plan_heavy(self, chunk: int) -> None:
        lens = np.diff(self.ptr_host.astype(np.int64))
        heavy = np.nonzero(lens > self.heavy_threshold)[0] if self.heavy_threshold > 0 else np.zeros(0, np.int64)
        self.n_heavy = int(heavy.shape[0])
        self.max_row_len = int(lens.max()) if lens.size else 0
        if self.n_heavy == 0:
            self.n_segments = 0
            self.heavy_rows = self.heavy_segptr = self.seg_begin = self.seg_end = None
            return
        # a row is cut into at most HEAVY_MAX_SEGMENTS pieces: the few extreme rows of a power-law graph (config C5: one query with millions of
        # hyperedges) get longer segments instead of tens of thousands of partial sums that ONE workgroup of the finish kernel adds up serially
        row_chunk = np.maximum(chunk, -(-lens[heavy] // HEAVY_MAX_SEGMENTS))
        row_chunk = row_chunk + (row_chunk & 1)                 # even: the two-hop list holds id PAIRS, a segment must hold whole pairs (ihg_node_pair_sums)
        seg_counts = (lens[heavy] + row_chunk - 1) // row_chunk
        segptr = np.zeros(self.n_heavy + 1, np.int64)
        np.cumsum(seg_counts, out=segptr[1:])
        self.n_segments = int(segptr[-1])
        owner = np.repeat(np.arange(self.n_heavy), seg_counts)
        within = np.arange(self.n_segments) - segptr[owner]
        begin = self.ptr_host[heavy][owner].astype(np.int64) + within * row_chunk[owner]
        end = np.minimum(begin + row_chunk[owner], self.ptr_host[heavy + 1][owner].astype(np.int64))
        dev = self.device
        self.heavy_rows = torch.from_numpy(heavy.astype(np.int32)).to(dev)
        self.heavy_segptr = torch.from_numpy(segptr.astype(np.int32)).to(dev)
        self.seg_begin = torch.from_numpy(begin.astype(np.int32)).to(dev)
        self.seg_end = torch.from_numpy(end.astype(np.int32)).to(dev)

    def _plan_order(self) -> None:
        """Rows by decreasing length (stable), heavy rows last: equal-length neighbours share a wave."""
        lens = np.diff(self.ptr_host.astype(np.int64))
        if self.n_rows < 2 or lens.max(initial=0) == lens.min(initial=0):
            self.row_order = None
            return
        key = lens.copy()
        if self.heavy_threshold > 0:
            key[lens > self.heavy_threshold] = -1          # skipped by the light kernel anyway
        order = np.argsort(-key, kind='stable').astype(np.int32)
        self.row_order = torch.from_numpy(order).to(self.device)

    def partials(self, dim: int) -> torch.Tensor:
        """Workspace for the split-row partial sums (allocated once per feature width)."""
        key = (self.n_segments, dim)
        buf = self._partials.get(key)
        if buf is None:
            buf = torch.empty(max(self.n_segments, 1), dim, dtype=torch.float32, device=self.device)
            self._partials[key] = buf
        return buf

    def row_slice(self, begin: int, end: int) -> 'CsrRows':
        return CsrRows(self, begin, end)

    def with_ids(self, ids_host: np.ndarray) -> 'Csr':
        """Same rows, row pointer and split-row plan, different id payload (shares every device array but ``ids``)."""
        other = Csr.__new__(Csr)
        other.__dict__.update(self.__dict__)
        other.ids_host = np.ascontiguousarray(ids_host, dtype=np.int32)
        other.ids = torch.from_numpy(other.ids_host).to(self.device)
        other._partials = self._partials
        return other

    def transpose(self, n_cols: int) -> 'Csr':
        lib = _lib.load()
        t_ptr = np.empty(n_cols + 1, np.int32)
        t_rows = np.empty(max(self.nnz, 1), np.int32)
        _lib.check(lib.ihg_transpose_csr(_as_ptr(self.ptr_host, ctypes.c_int32), _as_ptr(self.ids_host, ctypes.c_int32),
                                         self.n_rows, n_cols, _as_ptr(t_ptr, ctypes.c_int32), _as_ptr(t_rows, ctypes.c_int32)),
                   'ihg_transpose_csr')
        return Csr(t_ptr, t_rows[:self.nnz], self.device, self.heavy_threshold)


class CsrRows:
    """Rows [begin, end) of a :class:`Csr` (``ptr`` offsets are absolute, so a pointer bump suffices).

    Heavy rows inside the slice are processed in place by the light kernel (threshold disabled): slices are
    only used for the per-type passes of the interactive backward, whose source differs per node type.
    """

    def __init__(self, parent: Csr, begin: int, end: int):
        self.parent, self.begin, self.end = parent, begin, end
        self.n_rows = end - begin
        self.ptr = parent.ptr[begin:end + 1]
        self.ids = parent.ids
        self.device = parent.device
        self.heavy_threshold = 0
        self.n_heavy = 0
        self.n_segments = 0
        self.row_order = None


# IHG_EDGE_MULTIPLICITY = auto | 0 | 1: keep every DISTINCT (user, query, item) triple once and carry its number of occurrences as a per-hyperedge weight.  The reference
# makes one hyperedge per interaction, duplicates included (Helpers/Graph.py:107-118); m identical hyperedges contribute m times one hyperedge's value to every sum the path
# forms, so the collapsed layout computes the same function on E' <= E rows.  auto: on when at least MULTIPLICITY_MIN_SHARE of the interactions repeat an earlier triple
# of a graph of at least MULTIPLICITY_MIN_EDGES (config C5's power-law draw: 48.7 %; the C2 - C4 stand-ins: < 1 %, off - their kernels and bits stay what they were).
EDGE_MULTIPLICITY = os.environ.get('IHG_EDGE_MULTIPLICITY', 'auto')
MULTIPLICITY_MIN_SHARE = float(os.environ.get('IHG_MULTIPLICITY_MIN_SHARE', 0.25))
MULTIPLICITY_MIN_EDGES = int(os.environ.get('IHG_MULTIPLICITY_MIN_EDGES', 1 << 16))      # auto leaves small graphs alone (nothing to gain; '1' collapses any graph)
# the first-order launches walk the two-hop list with a row's repeated (destination, source) entries merged into one weighted entry when at least this share of
# the 6 E entries are repeats (C5: 79 %; C3 / C4 / C2: 8.6 / 13.6 / 17.9 %, where round 5 measured no gain - the repeats were cache hits); ops.TWO_HOP_MERGED overrides
TWO_HOP_MERGED_MIN_SHARE = float(os.environ.get('IHG_TWO_HOP_MERGED_MIN_SHARE', 0.25))
# IHG_COMPACT_NODES = auto | 0 | 1: number the nodes INSIDE the layout without the isolated ones (nodes that are in no hyperedge).  Every layer output of such a node is exactly
# zero (an empty sum times Dv^-1, App. B 2) and nobody gathers its rows, so the propagation - pair sums, node-level contractions, linear maps and their backward - runs on
# the N' nodes that have hyperedges; the public numbering (embedding tables, batch indices, evaluation, PpsHyperGraph's tensors) stays the reference's (Graph.py:110-111) and the
# two meet in RawGnn: X0 is gathered from the tables' active rows, the batch tail reads layer 0 from the tables and the layers above through the map (isolated: zero), the
# evaluation matrix is scattered back.  auto: on when at least COMPACT_MIN_SHARE of the nodes of a graph of >= COMPACT_MIN_NODES are isolated (config C5's power-law draw:
# 64 % - two thirds of every per-node kernel's rows; the C2 - C4 stand-ins: < 4 %, off).
COMPACT_NODES = os.environ.get('IHG_COMPACT_NODES', 'auto')
COMPACT_MIN_SHARE = float(os.environ.get('IHG_COMPACT_MIN_SHARE', 0.25))
COMPACT_MIN_NODES = int(os.environ.get('IHG_COMPACT_MIN_NODES', 1 << 16))


def unique_triples(triples: np.ndarray, user_count: int, query_count: int, item_count: int, count_only: bool = False):
    """``(distinct [E', 3] int64 ascending by (user, query, item), multiplicity [E'] float32, file_to_unique [E] int32)`` of an ``[E, 3]`` triple array
    (native ``ihg_unique_triples``); ``count_only``: just ``E'``."""
    lib = _lib.load()
    triples = np.ascontiguousarray(np.asarray(triples, dtype=np.int64).reshape(-1, 3))
    e = int(triples.shape[0])
    n = ctypes.c_int64(0)
    if count_only:
        _lib.check(lib.ihg_unique_triples(_as_ptr(triples, ctypes.c_int64), e, int(user_count), int(query_count), int(item_count), None, None, None, ctypes.byref(n)),
                   'ihg_unique_triples')
        return int(n.value)
    uniq = np.empty((max(e, 1), 3), np.int64)
    counts = np.empty(max(e, 1), np.float32)
    where = np.empty(max(e, 1), np.int32)
    _lib.check(lib.ihg_unique_triples(_as_ptr(triples, ctypes.c_int64), e, int(user_count), int(query_count), int(item_count), _as_ptr(uniq, ctypes.c_int64),
                                      _as_ptr(counts, ctypes.c_float), _as_ptr(where, ctypes.c_int32), ctypes.byref(n)), 'ihg_unique_triples')
    k = int(n.value)
    return np.ascontiguousarray(uniq[:k]), counts[:k].copy(), where[:e]


class IncidenceLayout:
    """The (user, query, item) hypergraph in kernel layout."""

    def __init__(self, triples: np.ndarray, user_count: int, query_count: int, item_count: int, device: torch.device,
                 heavy_threshold: Optional[int] = None, edge_order: str = os.environ.get('IHG_EDGE_ORDER', 'user'), edge_multiplicity: Optional[str] = None,
                 compact_nodes: Optional[str] = None):
        """``edge_order='user'`` renumbers the hyperedges by (user, file position) inside this layout: consecutive
        hyperedges then share their user row (the largest node table) and every user's incidence list is one contiguous
        run of edge-feature rows.  Hyperedge numbering is internal to the kernels - nothing outside the layout sees it
        (``triples_file_order`` keeps the caller's order, ``file_to_edge`` maps a file position to its row here); results are order-independent up to fp32
        re-association of the per-node sums.  ``edge_order='file'`` keeps the reference's numbering (Graph.py:107-118).

        ``edge_multiplicity`` (``'auto'`` | ``'0'`` | ``'1'``; default: ``IHG_EDGE_MULTIPLICITY``): collapse identical triples into ONE row of weight ``m_e``
        (``edge_weight``, ``None`` when off).  ``edge_count`` is then the number of DISTINCT hyperedges - the rows of every ``[E, d]`` buffer - while ``hyperedge_count``
        stays the reference's count (``PpsHyperGraph.EdgeCount``, the metric's E); degrees count every copy.  Rows are ordered by (user, query, item).

        ``compact_nodes`` (``'auto'`` | ``'0'`` | ``'1'``; default: ``IHG_COMPACT_NODES``): number the nodes inside the layout without the isolated ones.  ``user_count`` /
        ``query_count`` / ``item_count`` / ``node_count`` and every per-node array are then the COMPACT ones (what the kernels see); ``public_*`` are the reference's;
        ``node_map`` (``[N]`` int64: public global node id -> compact id, -1 for an isolated node) and ``active_nodes`` (``[N']``: compact -> public) translate, ``None`` when off."""
        lib = _lib.load()
        triples = np.ascontiguousarray(np.asarray(triples, dtype=np.int64).reshape(-1, 3))
        self.triples_file_order = triples
        self.hyperedge_count = int(triples.shape[0])
        self.public_user_count, self.public_query_count, self.public_item_count = int(user_count), int(query_count), int(item_count)
        self.public_node_count = self.public_user_count + self.public_query_count + self.public_item_count
        for m in range(3):
            if triples.shape[0] and (triples[:, m].min() < 0 or triples[:, m].max() >= (user_count, query_count, item_count)[m]):
                raise _lib.IhgnnHipError(f'hyperedge member {m} out of range [0, {(user_count, query_count, item_count)[m]})')
        # -- compact node numbering (decided on the public ids, before anything is built) --
        cmode = str(COMPACT_NODES if compact_nodes is None else compact_nodes)
        if cmode not in ('auto', '0', '1'):
            raise ValueError(f'compact_nodes: auto | 0 | 1, got {cmode!r}')
        self.node_map_host = self.active_nodes_host = None
        self.isolated_share = 0.0
        if cmode != '0' and triples.shape[0] > 0 and (cmode == '1' or self.public_node_count >= COMPACT_MIN_NODES):
            counts = (self.public_user_count, self.public_query_count, self.public_item_count)
            alive = []
            for m in range(3):
                seen = np.zeros(counts[m], bool)
                seen[triples[:, m]] = True
                alive.append(seen)
            n_alive = [int(a.sum()) for a in alive]
            self.isolated_share = 1.0 - sum(n_alive) / max(self.public_node_count, 1)
            if min(n_alive) > 0 and self.isolated_share > 0 and (cmode == '1' or self.isolated_share >= COMPACT_MIN_SHARE):
                node_map = np.full(self.public_node_count, -1, np.int64)
                active, pub0, cmp0 = [], 0, 0
                remapped = np.empty_like(triples)
                for m in range(3):
                    ids = np.nonzero(alive[m])[0]
                    local = np.full(counts[m], -1, np.int64)
                    local[ids] = np.arange(ids.shape[0])
                    remapped[:, m] = local[triples[:, m]]                # (monotone: an order by user stays an order by user)
                    node_map[pub0 + ids] = cmp0 + np.arange(ids.shape[0])
                    active.append(pub0 + ids)
                    pub0, cmp0 = pub0 + counts[m], cmp0 + ids.shape[0]
                self.node_map_host, self.active_nodes_host = node_map, np.concatenate(active)
                self._public_triples_for_views = triples
                triples = remapped
                user_count, query_count, item_count = n_alive
        self.user_count, self.query_count, self.item_count = int(user_count), int(query_count), int(item_count)
        self.node_count = n = self.user_count + self.query_count + self.item_count
        if edge_order not in ('user', 'file'):
            raise ValueError(f'unknown edge_order {edge_order!r}')
        mode = str(EDGE_MULTIPLICITY if edge_multiplicity is None else edge_multiplicity)
        if mode not in ('auto', '0', '1'):
            raise ValueError(f'edge_multiplicity: auto | 0 | 1, got {mode!r}')
        if mode == '1' and edge_order == 'file':
            raise ValueError('edge_multiplicity=1 renumbers the hyperedges by (user, query, item): it cannot keep edge_order="file"')
        multiplicity = None
        self.duplicate_share = 0.0
        if mode != '0' and edge_order == 'user' and self.hyperedge_count > 1 and (mode == '1' or self.hyperedge_count >= MULTIPLICITY_MIN_EDGES):
            distinct = unique_triples(triples, user_count, query_count, item_count, count_only=True)
            self.duplicate_share = 1.0 - distinct / self.hyperedge_count
            if mode == '1' or self.duplicate_share >= MULTIPLICITY_MIN_SHARE:
                triples, multiplicity, self.file_to_edge = unique_triples(triples, user_count, query_count, item_count)
                self.edge_perm = None
        if multiplicity is None:
            if edge_order == 'user' and triples.shape[0] > 1:
                self.edge_perm = np.argsort(triples[:, 0], kind='stable')        # new position -> file position
                triples = np.ascontiguousarray(triples[self.edge_perm])
                self.file_to_edge = np.empty(self.hyperedge_count, np.int32)
                self.file_to_edge[self.edge_perm] = np.arange(self.hyperedge_count, dtype=np.int32)
            else:
                self.edge_perm = None
                self.file_to_edge = np.arange(self.hyperedge_count, dtype=np.int32)
        self.edge_count = e = int(triples.shape[0])
        self.device = device
        self.node_map = None if self.node_map_host is None else torch.from_numpy(self.node_map_host).to(device)
        self.active_nodes = None if self.active_nodes_host is None else torch.from_numpy(self.active_nodes_host).to(device)
        i3 = np.empty((max(e, 1), 3), np.int32)
        rowptr = np.empty(n + 1, np.int32)
        edge_ids = np.empty(max(3 * e, 1), np.int32)
        degree = np.empty(max(n, 1), np.float32)
        _lib.check(lib.ihg_build_csr(_as_ptr(triples, ctypes.c_int64), e, self.user_count, self.query_count, self.item_count,
                                     _as_ptr(i3, ctypes.c_int32), _as_ptr(rowptr, ctypes.c_int32),
                                     _as_ptr(edge_ids, ctypes.c_int32), _as_ptr(degree, ctypes.c_float)),
                   'ihg_build_csr')
        self.i3_host = i3[:e]
        self.i3 = torch.from_numpy(self.i3_host).to(device)
        self.node_csr = Csr(rowptr, edge_ids[:3 * e], device, heavy_threshold)
        self.edge_weight_host = multiplicity
        self.edge_weight = None if multiplicity is None else torch.from_numpy(multiplicity).to(device)
        if multiplicity is not None:
            # degrees count every copy of a hyperedge (Graph.py:112: one increment per interaction); counts are exact in float32 below 2^24 like the reference's own
            counted = np.bincount(i3[:e].reshape(-1), weights=np.repeat(multiplicity.astype(np.float64), 3), minlength=n)[:n]
            degree[:n] = np.where(counted > 0, counted, 1e-8).astype(np.float32)
        deg = torch.from_numpy(degree[:n].copy())
        self.degree = deg.to(device)                       # Graph.py:120 semantics (isolated -> 1e-8)
        isolated = deg < 0.5
        # Same CPU float ops as the reference (GnnLayers.py:133,187) so the scale factors are bit-identical;
        # isolated nodes get 0 instead of 1e8 / 1e4: their sums are empty, so every output is unchanged.
        self.inv_deg = torch.where(isolated, torch.zeros_like(deg), deg.pow(-1)).to(device)
        self.inv_sqrt_deg = torch.where(isolated, torch.zeros_like(deg), deg.pow(-0.5)).to(device)
        u, q = self.user_count, self.query_count
        self.type_rows = (self.node_csr.row_slice(0, u), self.node_csr.row_slice(u, u + q), self.node_csr.row_slice(u + q, n))
        # member_csr: for node v and incident hyperedge e, the row of v's OWN slot in an [E,3,d] per-member buffer
        # (3e + type(v)); lets the interactive backward scatter-add all three member gradients in one K7 launch.
        lens = np.diff(rowptr.astype(np.int64))
        slot_of_node = np.zeros(n, np.int32)
        slot_of_node[u:u + q] = 1
        slot_of_node[u + q:] = 2
        slot_of_entry = np.repeat(slot_of_node, lens)
        self.member_csr = self.node_csr.with_ids(edge_ids[:3 * e] * 3 + slot_of_entry)
        # user-reduced form of the interactive backward (ihg_interact_bwd_user_reduced): the kernel sums the user slot on chip, so the
        # member buffer is [E, 2, d] (query, item) and only query / item nodes have lists: row of (node v, hyperedge e) = 2 e + type(v) - 1
        self.user_sorted = bool(e == 0 or (np.diff(i3[:e, 0]) >= 0).all())
        self._member_qi = None
        self._isolated_users = None
        self._rowptr_host, self._edge_ids_host, self._slot_of_entry = rowptr, edge_ids[:3 * e], slot_of_entry
        # hop2_csr: node v -> the OTHER two members of each of its hyperedges (2 ids per incidence); together with
        # self_weight = deg(v) for the node's own row it is the two-hop operator H H^T, read straight from the node table.
        inc = i3[:e][edge_ids[:3 * e]]                                   # [3E, 3] members of every incident hyperedge
        first = np.where(slot_of_entry == 0, 1, 0)
        second = np.where(slot_of_entry == 2, 1, 2)
        rows = np.arange(3 * e)
        others = np.stack([inc[rows, first], inc[rows, second]], axis=1).reshape(-1).astype(np.int32)
        self.hop2_csr = Csr((rowptr.astype(np.int64) * 2).astype(np.int32), others, device, heavy_threshold)
        # one weight per PAIR of the two-hop list (= per incidence): its hyperedge's multiplicity (ihg_node_pair_sums); None: every pair once
        self.pair_weight_host = None if multiplicity is None else multiplicity[edge_ids[:3 * e]]
        self.pair_weight = None if multiplicity is None else torch.from_numpy(self.pair_weight_host).to(device)
        self.self_weight = torch.where(isolated, torch.zeros_like(deg), deg).to(device)
        # share of the reference's 6 E two-hop entries that repeat a (destination, source) pair of their row (a count-only pass of the merge); the merged list itself
        # is built when a launch first asks for it
        nnz = ctypes.c_int64(0)
        merged_ptr = np.empty(self.hop2_csr.n_rows + 1, np.int32)
        _lib.check(lib.ihg_merge_id_lists(_as_ptr(self.hop2_csr.ptr_host, ctypes.c_int32), _as_ptr(self.hop2_csr.ids_host, ctypes.c_int32), None, self.hop2_csr.n_rows,
                                          _as_ptr(merged_ptr, ctypes.c_int32), None, None, ctypes.byref(nnz)), 'ihg_merge_id_lists')
        self.two_hop_duplicate_share = 1.0 - int(nnz.value) / max(6 * self.hyperedge_count, 1)
        # a layout with multiplicities has no unweighted two-hop list to fall back to: its first-order launches always walk the merged one
        self.two_hop_merged_default = bool(multiplicity is not None or self.two_hop_duplicate_share >= TWO_HOP_MERGED_MIN_SHARE)

    @property
    def compact(self) -> bool:
        """True when the layout numbers its nodes without the isolated ones (``node_map`` / ``active_nodes`` translate)."""
        return self.node_map is not None

    def public_degree(self) -> torch.Tensor:
        """``[N]`` vertex degrees in the PUBLIC numbering with the reference's 1e-8 for isolated nodes (``Graph.py:120``): ``degree`` itself unless the layout is compact."""
        if not self.compact:
            return self.degree
        cached = self.__dict__.get('_public_degree')
        if cached is None:
            cached = torch.full((self.public_node_count,), 1e-8, dtype=torch.float32, device=self.device)
            cached[self.active_nodes] = self.degree
            self.__dict__['_public_degree'] = cached
        return cached

    def compact_rows(self, public_rows: torch.Tensor, isolated_to: Optional[int] = None) -> torch.Tensor:
        """Public global node rows -> the layout's rows (int64; -1 for an isolated node, or ``isolated_to``); the rows themselves when the layout is not compact."""
        if not self.compact:
            return public_rows
        rows = self.node_map[public_rows]
        if isolated_to is not None:
            rows = torch.where(rows < 0, torch.full_like(rows, int(isolated_to)), rows)
        return rows

    def row_mask(self) -> torch.Tensor:
        """A byte per node row, all zero between uses (``ops._TwoHop.backward`` sets the rows of a sparse cotangent before its pull and clears them after)."""
        mask = self.__dict__.get('_row_mask')
        if mask is None:
            mask = self.__dict__['_row_mask'] = torch.zeros(self.node_count, dtype=torch.uint8, device=self.device)
        return mask

    def two_hop_merged(self) -> Tuple[Csr, torch.Tensor, float]:
        """``(csr, weights, duplicate_ratio)``: the two-hop list with the repeated (destination, source) entries of a row MERGED - the distinct other members of a
        node's hyperedges, ascending, and how often each occurs (``ihg_merge_id_lists``; with ``edge_weight``: the sum of the multiplicities).  ``H H^T - diag(deg)`` as
        a weighted CSR: the first-order layers' gather launches read one row per DISTINCT neighbour and scale it by the weight (the per-entry weight the K7 kernel
        already takes for ``Pps2DGraph``) - 8.6 % fewer gathers at C3, 13.6 % at C4, 17.9 % at C2, 79 % at C5 (a user meets the same query in several hyperedges).
        The pair sums of the interactive layer need the PAIRS and keep ``hop2_csr``.  ``duplicate_ratio`` is against the reference's 6 E entries.  Built on first
        use (a per-row sort of the list) - ``two_hop_duplicate_share`` is known from construction."""
        cached = self.__dict__.get('_two_hop_merged')
        if cached is None:
            src = self.hop2_csr
            ptr = np.empty(src.n_rows + 1, np.int32)
            ids = np.empty(max(src.nnz, 1), np.int32)
            counts = np.empty(max(src.nnz, 1), np.float32)
            nnz = ctypes.c_int64(0)
            weights = None if self.pair_weight_host is None else np.ascontiguousarray(np.repeat(self.pair_weight_host, 2))
            _lib.check(_lib.load().ihg_merge_id_lists(_as_ptr(src.ptr_host, ctypes.c_int32), _as_ptr(src.ids_host, ctypes.c_int32),
                                                      None if weights is None else _as_ptr(weights, ctypes.c_float), src.n_rows,
                                                      _as_ptr(ptr, ctypes.c_int32), _as_ptr(ids, ctypes.c_int32), _as_ptr(counts, ctypes.c_float), ctypes.byref(nnz)),
                       'ihg_merge_id_lists')
            n = int(nnz.value)
            csr = Csr(ptr, ids[:n].copy(), self.device, src.heavy_threshold)
            weights = torch.from_numpy(counts[:n].copy()).to(self.device)
            cached = self.__dict__['_two_hop_merged'] = (csr, weights, 1.0 - n / max(6 * self.hyperedge_count, 1))
        return cached

    def drop_row_mask(self) -> None:
        """Forget the mask (the next ``row_mask()`` makes a fresh all-zero one): what a user that raised between its set and its clear calls."""
        self.__dict__.pop('_row_mask', None)

    def users_without_hyperedges(self):
        """int64 device indices of the user rows with an empty incidence list (the user-reduced backward writes ``dh`` only for users that have
        hyperedges: these rows are zeroed by the caller - a handful of rows instead of a fill of the whole user block)."""
        if self._isolated_users is None:
            deg = np.diff(self._rowptr_host.astype(np.int64))[:self.user_count]
            self._isolated_users = torch.from_numpy(np.nonzero(deg == 0)[0].astype(np.int64)).to(self.device)
        return self._isolated_users

    def member_csr_qi(self):
        """``(Csr, rows)``: the member lists of the query and item nodes over an ``[E, 2, d]`` buffer (user rows are empty) and the
        int32 device list of those nodes, longest list first - what K7 walks after ``ihg_interact_bwd_user_reduced``."""
        if self._member_qi is None:
            u, n = self.user_count, self.node_count
            ptr = self._rowptr_host.astype(np.int64)
            first = ptr[u]
            ptr2 = np.zeros(n + 1, np.int64)
            ptr2[u:] = ptr[u:] - first
            ids2 = self._edge_ids_host[first:].astype(np.int64) * 2 + (self._slot_of_entry[first:].astype(np.int64) - 1)
            csr = Csr(ptr2.astype(np.int32), ids2.astype(np.int32), self.device, self.node_csr.heavy_threshold)
            lens = np.diff(ptr2)[u:]
            rows = (u + np.argsort(-lens, kind='stable')).astype(np.int32)
            self._member_qi = (csr, torch.from_numpy(rows).to(self.device))
        return self._member_qi

    def member_csr_qi_chunks(self, n_chunks: int):
        """``[(e0, e1, Csr, rows)]``: ``member_csr_qi`` cut by hyperedge range for the user-reduced backward in pieces (config C5 on one GPU: ``[E, 2, d]`` is
        102 GB), ids rebased to the chunk (``2 (e - e0) + slot - 1``).  The cuts fall where the USER changes (hyperedges are numbered by user: user ``u`` owns
        ``[rowptr[u], rowptr[u + 1])``), so no user's run of hyperedges meets two launches and every launch writes its own users' rows of ``dh``."""
        cache = self.__dict__.setdefault('_member_qi_chunks', {})
        if n_chunks not in cache:
            u, n, e = self.user_count, self.node_count, self.edge_count
            ptr = self._rowptr_host.astype(np.int64)
            starts = ptr[:u + 1]                                  # first hyperedge of every user (and E)
            step = -(-e // n_chunks)
            cuts = [0]
            for c in range(1, n_chunks):
                cut = int(starts[min(np.searchsorted(starts, c * step, side='left'), u)])
                if cut > cuts[-1] and cut < e:
                    cuts.append(cut)
            cuts.append(e)
            first = ptr[u]
            edge_of_entry = self._edge_ids_host[first:].astype(np.int64)
            slot = self._slot_of_entry[first:].astype(np.int64) - 1
            lens_all = np.diff(ptr)[u:]
            row_of_entry = np.repeat(np.arange(u, n, dtype=np.int64), lens_all)
            chunks = []
            for e0, e1 in zip(cuts[:-1], cuts[1:]):
                pick = (edge_of_entry >= e0) & (edge_of_entry < e1)
                counts = np.bincount(row_of_entry[pick], minlength=n)
                ptr2 = np.zeros(n + 1, np.int64)
                np.cumsum(counts, out=ptr2[1:])
                ids2 = (edge_of_entry[pick] - e0) * 2 + slot[pick]
                csr = Csr(ptr2.astype(np.int32), ids2.astype(np.int32), self.device, self.node_csr.heavy_threshold)
                rows = (u + np.argsort(-counts[u:], kind='stable')).astype(np.int32)
                chunks.append((e0, e1, csr, torch.from_numpy(rows).to(self.device)))
            cache[n_chunks] = chunks
        return cache[n_chunks]

    def member_csr_chunks(self, n_chunks: int):
        """``[(e0, e1, Csr)]``: the member lists of ``member_csr`` cut by hyperedge range, ids rebased to the chunk
        (``3 (e - e0) + type(v)``).  For the interactive backward when the ``[E, 3, d]`` member-gradient buffer has to be produced
        in pieces (config C5 on one GPU); built on first use and kept."""
        cache = self.__dict__.setdefault('_member_chunks', {})
        if n_chunks not in cache:
            e = self.edge_count
            step = -(-e // n_chunks)
            step = -(-step // 128) * 128                     # whole tiles of either tile size
            mid = self.member_csr.ids_host.astype(np.int64)
            lens = np.diff(self.member_csr.ptr_host.astype(np.int64))
            row_of_entry = np.repeat(np.arange(self.node_count, dtype=np.int64), lens)
            chunks = []
            for e0 in range(0, e, step):
                e1 = min(e0 + step, e)
                pick = (mid >= 3 * e0) & (mid < 3 * e1)
                counts = np.bincount(row_of_entry[pick], minlength=self.node_count)
                ptr = np.zeros(self.node_count + 1, np.int64)
                np.cumsum(counts, out=ptr[1:])
                chunks.append((e0, e1, Csr(ptr.astype(np.int32), (mid[pick] - 3 * e0).astype(np.int32), self.device, self.member_csr.heavy_threshold)))
            cache[n_chunks] = chunks
        return cache[n_chunks]


class LogHyperLayout:
    """General hypergraph incidence in kernel layout: hyperedges of any arity, both orientations as CSR with per-entry values
    (edge-major: hyperedge -> member nodes, the node->hyperedge gather; node-major: node -> incident hyperedges, the
    hyperedge->node gather).  Both run on the K7 segment-sum kernel, whose split-row plan and length-sorted row order are the
    degree-bucketed treatment of the long member / incidence lists.  Built by the native ``ihg_build_log_hypergraph``."""

    def __init__(self, triples: np.ndarray, pos_log: np.ndarray, user_count: int, query_count: int, item_count: int, device: torch.device,
                 heavy_threshold: Optional[int] = None):
        lib = _lib.load()
        triples = np.ascontiguousarray(np.asarray(triples, dtype=np.int64).reshape(-1, 3))
        pos_log = np.ascontiguousarray(np.asarray(pos_log, dtype=np.int64).reshape(-1))
        if pos_log.shape[0] != triples.shape[0]:
            raise ValueError('one search-log row per positive is required')
        p = int(triples.shape[0])
        self.user_count, self.query_count, self.item_count = int(user_count), int(query_count), int(item_count)
        self.node_count = n = self.user_count + self.query_count + self.item_count
        self.device = device
        edge_ptr = np.empty(p + 1, np.int32)
        edge_nodes = np.empty(max(3 * p, 1), np.int32)
        edge_vals = np.empty(max(3 * p, 1), np.float32)
        edge_degree = np.empty(max(p, 1), np.float32)
        node_ptr = np.empty(n + 1, np.int32)
        node_edges = np.empty(max(3 * p, 1), np.int32)
        node_vals = np.empty(max(3 * p, 1), np.float32)
        node_degree = np.empty(max(n, 1), np.float32)
        n_edges, nnz = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(lib.ihg_build_log_hypergraph(_as_ptr(triples, ctypes.c_int64), _as_ptr(pos_log, ctypes.c_int64), p, self.user_count,
                                                self.query_count, self.item_count, _as_ptr(edge_ptr, ctypes.c_int32),
                                                _as_ptr(edge_nodes, ctypes.c_int32), _as_ptr(edge_vals, ctypes.c_float),
                                                _as_ptr(edge_degree, ctypes.c_float), _as_ptr(node_ptr, ctypes.c_int32),
                                                _as_ptr(node_edges, ctypes.c_int32), _as_ptr(node_vals, ctypes.c_float),
                                                _as_ptr(node_degree, ctypes.c_float), ctypes.byref(n_edges), ctypes.byref(nnz)),
                   'ihg_build_log_hypergraph')
        e, k = int(n_edges.value), int(nnz.value)
        self.edge_count, self.nnz = e, k
        self.edge_csr = Csr(edge_ptr[:e + 1], edge_nodes[:k], device, heavy_threshold)          # rows = hyperedges, ids = member nodes
        self.node_csr = Csr(node_ptr, node_edges[:k], device, heavy_threshold)                    # rows = nodes, ids = hyperedges
        self.edge_values_host, self.node_values_host = edge_vals[:k].copy(), node_vals[:k].copy()
        unit = bool((self.edge_values_host == 1.0).all())
        self.edge_values = None if unit else torch.from_numpy(self.edge_values_host).to(device)   # unit incidence: no value stream
        self.node_values = None if unit else torch.from_numpy(self.node_values_host).to(device)
        deg = torch.from_numpy(node_degree[:n].copy())
        self.degree = deg.to(device)
        isolated = deg < 0.5
        self.inv_sqrt_deg = torch.where(isolated, torch.zeros_like(deg), deg.pow(-0.5)).to(device)       # GnnLayers.py:133
        edeg = torch.from_numpy(edge_degree[:e].copy())
        self.edge_degree = edeg.to(device)
        self.inv_edge_degree = edeg.pow(-1).to(device)                                                    # GnnLayers.py:134


COMPLETENESS = {'uqi': 0, 'uq': 1, 'ui': 2, 'qi': 3}


class PairLayout:
    """Pairwise (user-query-item) graph of the GCN baseline in kernel layout: symmetric weighted CSR + ``D^-1/2``."""

    def __init__(self, triples: np.ndarray, user_count: int, query_count: int, item_count: int, device: torch.device,
                 completeness: str = 'uqi', self_loops: bool = False, heavy_threshold: Optional[int] = None):
        lib = _lib.load()
        if completeness not in COMPLETENESS:
            raise ValueError(f'unknown graph completeness {completeness!r}')
        triples = np.ascontiguousarray(np.asarray(triples, dtype=np.int64).reshape(-1, 3))
        e = int(triples.shape[0])
        self.node_count = n = int(user_count + query_count + item_count)
        self.user_count, self.query_count, self.item_count = int(user_count), int(query_count), int(item_count)
        cap = 6 * e + n + 1
        rowptr = np.empty(n + 1, np.int32)
        cols = np.empty(cap, np.int32)
        vals = np.empty(cap, np.float32)
        degree = np.empty(max(n, 1), np.float32)
        nnz = ctypes.c_int64(0)
        _lib.check(lib.ihg_build_pair_csr(_as_ptr(triples, ctypes.c_int64), e, self.user_count, self.query_count, self.item_count,
                                          COMPLETENESS[completeness], int(bool(self_loops)), _as_ptr(rowptr, ctypes.c_int32),
                                          _as_ptr(cols, ctypes.c_int32), _as_ptr(vals, ctypes.c_float), _as_ptr(degree, ctypes.c_float),
                                          cap, ctypes.byref(nnz)), 'ihg_build_pair_csr')
        k = int(nnz.value)
        self.device = device
        self.csr = Csr(rowptr, cols[:k], device, heavy_threshold)
        self.values_host = vals[:k].copy()
        self.values = torch.from_numpy(self.values_host).to(device)
        deg = torch.from_numpy(degree[:n].copy())
        self.degree = deg.to(device)
        isolated = deg < 0.5
        self.inv_sqrt_deg = torch.where(isolated, torch.zeros_like(deg), deg.pow(-0.5)).to(device)
