"""Graph objects handed from the dataset to the GNN layers (reference ``Helpers/Graph.py``).

``PpsHyperGraph`` keeps the reference's attribute surface (``Adjacency``, ``I3``, ``VertexDegrees``,
``EdgeDegrees``, ``EdgeCount``; ``Graph.py:84-92``) but what the layers actually consume is ``layout``: the
int32 edge-major / node-major-CSR incidence built natively by ``ihg_build_csr`` (one counting sort on the
host, instead of the reference's Python loop with one device op per interaction, ``Graph.py:107-118``).
The reference-shaped tensors are materialised lazily, on first access, for callers that still want them.
"""
from typing import Iterable, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from ..layout import IncidenceLayout, LogHyperLayout


class PpsGraph:
    """Base class of the graph containers."""


class PpsHyperGraph(PpsGraph):
    """One hyperedge per positive (user, query, item) interaction; duplicates stay distinct hyperedges."""

    def __init__(self):
        self.layout: Optional[IncidenceLayout] = None
        self.EdgeCount = 0
        self._adjacency = self._i3_long = self._edge_degrees = None

    @classmethod
    def from_triples(cls, triples: np.ndarray, node_count: int, user_count: int, query_count: int,
                     device: torch.device) -> 'PpsHyperGraph':
        """Build from an ``[E,3]`` array of 0-based per-type (user, query, item) ids."""
        g = cls()
        item_count = node_count - user_count - query_count
        g.layout = IncidenceLayout(triples, user_count, query_count, item_count, device)
        g.EdgeCount = g.layout.hyperedge_count            # the reference's count: one hyperedge per positive interaction, duplicates included (Graph.py:133)
        return g

    @classmethod
    def from_interactions(cls, interactions: Iterable, node_count: int, user_count: int, query_count: int,
                          device: torch.device) -> 'PpsHyperGraph':
        """Reference entry point (``Graph.py:94-100``): ``interactions`` yield ``.uqif()`` tuples; flag <= 0 is skipped."""
        rows = [(u, q, i) for u, q, i, flag in (p.uqif() for p in interactions) if flag > 0]
        return cls.from_triples(np.asarray(rows, dtype=np.int64).reshape(-1, 3), node_count, user_count, query_count, device)

    # -- reference-shaped views ---------------------------------------------------------------
    @property
    def VertexDegrees(self) -> Tensor:
        """``[N,1]`` float degrees with isolated nodes at 1e-8 (``Graph.py:120,131``), in the reference's node numbering (the layout's own may leave the isolated nodes out)."""
        return self.layout.public_degree().view(-1, 1)

    @property
    def EdgeDegrees(self) -> Tensor:
        if self._edge_degrees is None:
            self._edge_degrees = torch.full((self.EdgeCount, 1), 3.0, dtype=torch.float32, device=self.layout.device)
        return self._edge_degrees

    @property
    def I3(self) -> Tensor:
        """``[E,3]`` int64 global member ids in FILE order (``Graph.py:129``); the kernels' own copy is ``layout.i3``."""
        if self._i3_long is None:
            lay = self.layout
            offs = np.array([0, lay.public_user_count, lay.public_user_count + lay.public_query_count], dtype=np.int64)
            self._i3_long = torch.from_numpy(lay.triples_file_order + offs).to(lay.device)
        return self._i3_long

    @property
    def Adjacency(self) -> Tensor:
        """Coalesced ``[N x E]`` unit-valued sparse COO incidence with file-order hyperedge numbering (``Graph.py:123-128``)."""
        if self._adjacency is None:
            i3 = self.I3.cpu()
            rows = i3.reshape(-1)
            cols = torch.arange(self.EdgeCount, dtype=torch.int64).repeat_interleave(3)
            adj = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.ones(3 * self.EdgeCount, dtype=torch.float32),
                                          (self.layout.public_node_count, self.EdgeCount)).coalesce()
            self._adjacency = adj.to(self.layout.device)
        return self._adjacency


class PpsLogHyperGraph(PpsGraph):
    """One hyperedge of VARIABLE arity per search log with at least one positive: members = user, query and every positive item
    of the log (``Graph.py:138-189``).  Kernel layout in ``layout`` (edge-major and node-major CSR with values, from the native
    ``ihg_build_log_hypergraph``); ``Adjacency`` / ``VertexDegrees`` / ``EdgeDegrees`` are the reference-shaped views.  Only
    ``HGCNLayer`` runs on it (it has no ``I3``, which ``IHGNNLayer`` needs - as in the reference)."""

    def __init__(self):
        self.layout: Optional[LogHyperLayout] = None
        self.EdgeCount = 0
        self._adjacency = None

    @classmethod
    def from_positives(cls, triples: np.ndarray, pos_log: np.ndarray, node_count: int, user_count: int, query_count: int,
                       device: torch.device) -> 'PpsLogHyperGraph':
        """``triples [P,3]`` 0-based (user, query, item) positives in file order, ``pos_log [P]`` their search-log row."""
        g = cls()
        g.layout = LogHyperLayout(triples, pos_log, user_count, query_count, node_count - user_count - query_count, device)
        g.EdgeCount = g.layout.edge_count
        return g

    @classmethod
    def from_search_logs(cls, logs: Iterable, node_count: int, user_count: int, query_count: int, device: torch.device) -> 'PpsLogHyperGraph':
        """Reference entry point (``Graph.py:149-155``): ``logs`` yield objects with ``user``, ``query``, ``items``, ``interactions``."""
        rows, owner = [], []
        for k, log in enumerate(logs):
            for item, flag in zip(log.items, log.interactions):
                if flag > 0:
                    rows.append((log.user, log.query, item))
                    owner.append(k)
        return cls.from_positives(np.asarray(rows, dtype=np.int64).reshape(-1, 3), np.asarray(owner, dtype=np.int64), node_count,
                                  user_count, query_count, device)

    @property
    def VertexDegrees(self) -> Tensor:
        return self.layout.degree.view(-1, 1)

    @property
    def EdgeDegrees(self) -> Tensor:
        return self.layout.edge_degree.view(-1, 1)

    @property
    def Adjacency(self) -> Tensor:
        """Coalesced ``[N x E]`` sparse COO incidence (``Graph.py:178-184``; a repeated item of a log has value 2)."""
        if self._adjacency is None:
            csr = self.layout.node_csr
            lens = np.diff(csr.ptr_host.astype(np.int64))
            rows = torch.from_numpy(np.repeat(np.arange(csr.n_rows, dtype=np.int64), lens))
            cols = torch.from_numpy(csr.ids_host.astype(np.int64))
            self._adjacency = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.from_numpy(self.layout.node_values_host),
                                                      (self.layout.node_count, self.EdgeCount)).coalesce().to(self.layout.device)
        return self._adjacency


class Pps2DGraph(PpsGraph):
    """Pairwise graph of the GCN baseline (``Graph.py:13-81``): every positive interaction links its members pairwise
    (which pairs: ``Gs.graph_completeness``), both directions, duplicate links summed.  Kernel layout in ``layout``
    (symmetric weighted CSR from the native ``ihg_build_pair_csr``); ``Adjacency`` / ``VertexDegrees`` are the
    reference-shaped views."""

    def __init__(self):
        self.layout = None
        self._adjacency = None

    @classmethod
    def from_triples(cls, triples: np.ndarray, node_count: int, user_count: int, query_count: int, use_self_connection: bool,
                     device: torch.device, completeness: Optional[str] = None) -> 'Pps2DGraph':
        from ..layout import PairLayout
        from .GlobalSettings import Gs
        g = cls()
        g.layout = PairLayout(triples, user_count, query_count, node_count - user_count - query_count, device,
                              completeness or Gs.graph_completeness, use_self_connection)
        return g

    @classmethod
    def from_interactions(cls, interactions: Iterable, node_count: int, user_count: int, query_count: int,
                          use_self_connection: bool, device: torch.device) -> 'Pps2DGraph':
        rows = [(u, q, i) for u, q, i, flag in (p.uqif() for p in interactions) if flag > 0]
        return cls.from_triples(np.asarray(rows, dtype=np.int64).reshape(-1, 3), node_count, user_count, query_count,
                                use_self_connection, device)

    @property
    def VertexDegrees(self) -> Tensor:
        return self.layout.degree.view(-1, 1)

    @property
    def Adjacency(self) -> Tensor:
        if self._adjacency is None:
            csr = self.layout.csr
            lens = np.diff(csr.ptr_host.astype(np.int64))
            rows = torch.from_numpy(np.repeat(np.arange(csr.n_rows, dtype=np.int64), lens))
            cols = torch.from_numpy(csr.ids_host.astype(np.int64))
            n = self.layout.node_count
            self._adjacency = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.from_numpy(self.layout.values_host),
                                                      (n, n)).coalesce().to(self.layout.device)
        return self._adjacency
