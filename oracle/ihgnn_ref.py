"""CPU oracle for the IHGNN hypergraph message-passing path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``ihgnn_amd``) never does and fails loudly without its HIP library.

It restates, in plain PyTorch-CPU ops, the exact ATen op sequence the reference issues on the path
(SURVEY.md §2b K1-K13), so it doubles as the "reference PyTorch-CPU path" timed beside the GPU.
Each function cites the reference lines it follows (paths relative to the reference checkout).

PINNING: every function here is checked in ``tests/test_oracle_golden.py`` against fixtures produced
by running the real reference on CPU (``tests/golden/make_golden.py``; layer outputs and gradients,
model scores, one Adam step, a 48-step loss curve, ranking metrics, graph tensors).  The one piece of
third-party arithmetic on the path, ``torch_sparse.matmul`` (unpinned upstream, not in the image), is
restated as ``torch.sparse.mm`` on the coalesced unit-valued incidence - the published semantics of
SpMM with sum reduction - both here and in the fixture generator's stub.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor


# ---------------------------------------------------------------------------------------------
# Graph (Helpers/Graph.py:94-134)
# ---------------------------------------------------------------------------------------------
class HyperGraph:
    """(user, query, item) hypergraph tensors as ``PpsHyperGraph.from_interactions`` builds them."""

    def __init__(self, triples: np.ndarray, user_count: int, query_count: int, item_count: int,
                 dtype: torch.dtype = torch.float32):
        triples = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
        self.user_count, self.query_count, self.item_count = user_count, query_count, item_count
        self.node_count = n = user_count + query_count + item_count
        self.edge_count = e = triples.shape[0]
        # Graph.py:110-111: global ids u, q+U, i+U+Q
        i3 = triples + np.array([0, user_count, user_count + query_count], dtype=np.int64)
        self.I3 = torch.from_numpy(i3.copy())
        # Graph.py:112,120: degree = incident hyperedges, zeros -> 1e-8
        deg = np.bincount(i3.reshape(-1), minlength=n).astype(np.float64)
        deg[deg == 0] = 1e-8
        self.VertexDegrees = torch.from_numpy(deg).to(dtype).view(-1, 1)
        self.EdgeDegrees = torch.full((e, 1), 3.0, dtype=dtype)           # Graph.py:132
        # Graph.py:123-128: coalesced COO [N x E], unit values
        rows = torch.from_numpy(i3.reshape(-1))
        cols = torch.arange(e, dtype=torch.int64).repeat_interleave(3)
        self.Adjacency = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.ones(3 * e, dtype=dtype),
                                                 (n, e)).coalesce()
        self.AdjacencyT = self.Adjacency.t().coalesce()

    def to(self, dtype: torch.dtype) -> 'HyperGraph':
        g = HyperGraph.__new__(HyperGraph)
        g.__dict__.update(self.__dict__)
        g.VertexDegrees = self.VertexDegrees.to(dtype)
        g.EdgeDegrees = self.EdgeDegrees.to(dtype)
        g.Adjacency = self.Adjacency.to(dtype)
        g.AdjacencyT = self.AdjacencyT.to(dtype)
        return g


class LogHyperGraph:
    """Per-search-log hypergraph as ``PpsLogHyperGraph.from_search_logs`` builds it (Helpers/Graph.py:138-189): one hyperedge per log
    with >= 1 positive, members [user, query + U, positive items + U + Q]; ``logs`` = iterable of (user, query, items, flags)."""

    def __init__(self, logs, user_count: int, query_count: int, item_count: int):
        self.user_count, self.query_count, self.item_count = user_count, query_count, item_count
        self.node_count = n = user_count + query_count + item_count
        rows, cols, edge_degrees = [], [], []
        deg = np.zeros(n, dtype=np.float64)
        edge = 0
        for u, q, items, flags in logs:
            nodes = [u, q + user_count] + [i + user_count + query_count for i, f in zip(items, flags) if f > 0]   # Graph.py:160-162
            if len(nodes) == 2:                                                                                   # :163
                continue
            deg[np.array(nodes)] += 1                    # :166  (fancy-index += : once per distinct node)
            edge_degrees.append(len(nodes))              # :167
            rows += nodes
            cols += [edge] * len(nodes)
            edge += 1
        deg[deg == 0] = 1e-8                             # :171
        self.edge_count = edge
        idx = torch.tensor([rows, cols], dtype=torch.int64) if rows else torch.zeros(2, 0, dtype=torch.int64)
        self.Adjacency = torch.sparse_coo_tensor(idx, torch.ones(len(rows)), (n, edge)).coalesce()      # :174-181: duplicates summed
        self.AdjacencyT = self.Adjacency.t().coalesce()
        self.VertexDegrees = torch.from_numpy(deg).float().view(-1, 1)
        self.EdgeDegrees = torch.tensor(edge_degrees, dtype=torch.float32).view(-1, 1)


class PairGraph:
    """Pairwise graph as ``Pps2DGraph.from_interactions`` builds it (Helpers/Graph.py:19-81), no self connections."""

    PAIRS = {'uqi': ((0, 1), (1, 2), (2, 0)), 'uq': ((0, 1),), 'ui': ((0, 2),), 'qi': ((1, 2),)}

    def __init__(self, triples: np.ndarray, user_count: int, query_count: int, item_count: int, completeness: str = 'uqi'):
        triples = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
        self.user_count, self.query_count, self.item_count = user_count, query_count, item_count
        self.node_count = n = user_count + query_count + item_count
        nodes = triples + np.array([0, user_count, user_count + query_count], dtype=np.int64)       # Graph.py:38-39
        rows, cols = [], []
        deg = np.zeros(n, dtype=np.float64)
        for a, b in self.PAIRS[completeness]:                                                        # Graph.py:40-63
            rows += [nodes[:, a], nodes[:, b]]
            cols += [nodes[:, b], nodes[:, a]]
            np.add.at(deg, nodes[:, a], 1.0)
            np.add.at(deg, nodes[:, b], 1.0)
        deg[deg == 0] = 1e-8                                                                          # Graph.py:68-69
        idx = torch.from_numpy(np.stack([np.concatenate(rows), np.concatenate(cols)])) if len(triples) else torch.zeros(2, 0, dtype=torch.long)
        self.Adjacency = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (n, n)).coalesce()   # Graph.py:72-78: duplicates summed
        self.VertexDegrees = torch.from_numpy(deg).float().view(-1, 1)


# ---------------------------------------------------------------------------------------------
# Embeddings (Models/EmbeddingLayers.py:51-91)
# ---------------------------------------------------------------------------------------------
def embed_all_nodes(w_user: Tensor, w_vocab: Tensor, w_item: Tensor, bag_input: Tensor, bag_offsets: Tensor) -> Tensor:
    """X0 = [Wu[1..U]; mean-bag over query words; Wi[1..I]]  (EmbeddingLayers.py:70-79, RawGnn.py:112).

    ``bag_input`` holds word ids already shifted by +1 (Dataset.py:168).
    """
    users = F.embedding(torch.arange(1, w_user.shape[0]), w_user)
    items = F.embedding(torch.arange(1, w_item.shape[0]), w_item)
    queries = F.embedding_bag(bag_input, w_vocab, bag_offsets, mode='mean')
    return torch.cat([users, queries, items])


# ---------------------------------------------------------------------------------------------
# Layers
# ---------------------------------------------------------------------------------------------
def feature_interactor(h: Tensor, i3: Tensor, weight: Tensor, bias: Tensor, order: int) -> Tensor:
    """node -> hyperedge (Models/CommonLayers.py:58-87)."""
    if order == 1:
        sel = h[i3]                                              # [E,3,d]        :62
        return F.linear(sel.reshape(-1, 3 * h.shape[1]), weight, bias)   # :64-66
    u, q, i = h[i3[:, 0]], h[i3[:, 1]], h[i3[:, 2]]              # :70-72
    uq, qi, iu = u * q, q * i, i * u                             # :74-76
    blocks = [u, q, i, uq, qi, iu]
    if order == 3:
        blocks.append(uq * i)                                    # :79
    return F.linear(torch.cat(blocks, 1), weight, bias)          # :81-85


def ihgnn_layer(x: Tensor, g: HyperGraph, wt: Tensor, bt: Tensor, wa: Tensor, ba: Tensor, order: int) -> Tensor:
    """IHGNNLayer.forward, phase2_attention=False (Models/GnnLayers.py:221-236)."""
    h = F.linear(x, wt, bt)                                      # :224
    ef = feature_interactor(h, g.I3, wa, ba, order)              # :225
    y = torch.sparse.mm(g.Adjacency, ef)                         # :233
    return g.VertexDegrees.pow(-1) * y                           # :187,:234


def hgcn_layer(x: Tensor, g: HyperGraph, w: Tensor, b: Tensor) -> Tensor:
    """HGCNLayer.forward (Models/GnnLayers.py:142-153)."""
    dv = g.VertexDegrees.pow(-0.5)                               # :133
    h = dv * F.linear(x, w, b)                                   # :145-146
    ef = g.EdgeDegrees.pow(-1) * torch.sparse.mm(g.AdjacencyT, h)   # :148-149
    return dv * torch.sparse.mm(g.Adjacency, ef)                 # :151-152


def gcn_layer(x: Tensor, g: PairGraph, w: Tensor, b: Tensor) -> Tensor:
    """GCNLayer.forward (Models/GnnLayers.py:28-45)."""
    dv = g.VertexDegrees.pow(-0.5)                               # :24
    if w.shape[1] >= w.shape[0]:                                 # :33-37  transform first when it does not widen
        return dv * torch.sparse.mm(g.Adjacency, dv * F.linear(x, w, b))
    return F.linear(dv * torch.sparse.mm(g.Adjacency, dv * x), w, b)     # :38-43


def hem_score(user_f: Tensor, query_f: Tensor, item_f: Tensor, items_bias: Tensor, lam: float) -> Tensor:
    """HemPredictionLayer.forward (Models/PredictionLayers.py:21-44), dot-product branch."""
    m_uq = lam * query_f + (1 - lam) * user_f                    # :35
    return (item_f * m_uq).sum(1) + items_bias                   # :42-43


class OracleRawGnn(nn.Module):
    """RawGnn restated (Models/RawGnn.py:14-158) with the reference's state-dict key names."""

    def __init__(self, g: HyperGraph, bag_input: Tensor, bag_offsets: Tensor, vocab_size: int, dim: int,
                 layer_kind: str, layer_count: int, order: int, lam: float = 0.5, dtype: torch.dtype = torch.float32):
        super().__init__()
        assert layer_kind in ('ihgnn', 'hgcn', 'gcn')
        self.g, self.bag_input, self.bag_offsets = g, bag_input, bag_offsets
        self.kind, self.layer_count, self.lam, self.dim = layer_kind, layer_count, lam, dim
        # RawGnn.py:76-78: only layer 0 keeps the requested interaction order
        self.orders = [order if (l == 0 or order == 1) else 1 for l in range(layer_count)]
        shapes = {
            'embeddings.embedding_user.weight': (g.user_count + 1, dim),
            'embeddings.embedding_item.weight': (g.item_count + 1, dim),
            'embeddings.embedding_bag_vocabulary.weight': (vocab_size + 1, dim),
            'prediction_layer.items_bias': (g.item_count,),
        }
        for l in range(layer_count):
            shapes[f'gnn_{l}.feature_transform.weight'] = (dim, dim)
            shapes[f'gnn_{l}.feature_transform.bias'] = (dim,)
            if layer_kind == 'ihgnn':
                k = {1: 3, 2: 6, 3: 7}[self.orders[l]]
                shapes[f'gnn_{l}.feature_interactor.aggregation.weight'] = (dim, k * dim)
                shapes[f'gnn_{l}.feature_interactor.aggregation.bias'] = (dim,)
        self.key_of = {}
        for key, shape in shapes.items():
            attr = key.replace('.', '__')
            self.key_of[attr] = key
            self.register_parameter(attr, nn.Parameter(torch.zeros(shape, dtype=dtype)))
        self._saved: Optional[Tensor] = None
        self.pair_graph: Optional[PairGraph] = None            # set by the caller for layer_kind == 'gcn'

    # -- state-dict in the reference's key space -------------------------------------------
    def load_reference_state(self, sd: Dict[str, np.ndarray]) -> None:
        with torch.no_grad():
            for attr, key in self.key_of.items():
                p = getattr(self, attr)
                p.copy_(torch.as_tensor(np.asarray(sd[key])).to(p.dtype))

    def reference_state(self) -> Dict[str, Tensor]:
        return {key: getattr(self, attr).detach().clone() for attr, key in self.key_of.items()}

    def reference_grads(self) -> Dict[str, Tensor]:
        return {key: getattr(self, attr).grad.detach().clone() for attr, key in self.key_of.items()}

    def P(self, key: str) -> Tensor:
        return getattr(self, key.replace('.', '__'))

    # -- RawGnn.py:110-122 --------------------------------------------------------------------
    def propagate(self) -> Tensor:
        x = embed_all_nodes(self.P('embeddings.embedding_user.weight'), self.P('embeddings.embedding_bag_vocabulary.weight'),
                            self.P('embeddings.embedding_item.weight'), self.bag_input, self.bag_offsets)
        outs = [x]
        for l in range(self.layer_count):
            wt, bt = self.P(f'gnn_{l}.feature_transform.weight'), self.P(f'gnn_{l}.feature_transform.bias')
            if self.kind == 'ihgnn':
                x = ihgnn_layer(x, self.g, wt, bt, self.P(f'gnn_{l}.feature_interactor.aggregation.weight'),
                                self.P(f'gnn_{l}.feature_interactor.aggregation.bias'), self.orders[l])
            elif self.kind == 'hgcn':
                x = hgcn_layer(x, self.g, wt, bt)
            else:
                x = gcn_layer(x, self.pair_graph, wt, bt)
            outs.append(x)
        return torch.cat(outs, 1)

    def forward(self, users: Tensor, queries: Tensor, items: Optional[Tensor] = None) -> Tensor:
        feats = self.propagate() if self._saved is None else self._saved
        g = self.g
        uf = feats[users]                                             # RawGnn.py:128
        qf = feats[queries + g.user_count]                            # :129
        if items is not None:
            itf = feats[items + g.user_count + g.query_count]         # :131
            bias = self.P('prediction_layer.items_bias')[items]
        else:
            itf = feats[g.user_count + g.query_count:]                # :133
            bias = self.P('prediction_layer.items_bias')
        return hem_score(uf, qf, itf, bias, self.lam)

    def save_features_for_test(self) -> None:                         # RawGnn.py:147-155
        self._saved = self.propagate()

    def clear_saved_feature(self) -> None:
        self._saved = None


# ---------------------------------------------------------------------------------------------
# Ranking metrics (Helpers/Metrics.py:46-109)
# ---------------------------------------------------------------------------------------------
def ranking_metrics(scores: Tensor, truth: Sequence[int], flags: Optional[Sequence[int]] = None) -> Tuple[float, float, float]:
    """(HR@10, NDCG@10, MAP@10) of one search over all items; ``flags=None`` = all relevance 1."""
    top = torch.sort(scores, descending=True)[1][:10].tolist()                  # :60-61
    cap = min(len(truth), 10)                                                     # :63
    if flags is None:
        hits = [top.index(t) for t in truth if t in top]                          # :66-69
        dcg = sum(math.log(2, r + 2) for r in hits)                               # :96
        idcg = sum(math.log(2, r) for r in range(2, 2 + cap))                     # :100-107
    else:
        pairs = [(top.index(t), f) for t, f in zip(truth, flags) if t in top]     # :71-77
        hits = [p for p, _ in pairs]
        dcg = sum(math.log(2, r + 2) * (2 ** f - 1) for r, f in pairs)            # :94
        idcg = sum(math.log(2, r + 2) * (2 ** f - 1)
                   for r, f in enumerate(sorted((f for _, f in pairs), reverse=True)))   # :98
    hr = len(hits) / cap                                                          # :80
    ap = 0.0 if not hits else sum(j / (r + 1) for r, j in zip(hits, range(1, len(hits) + 1))) / len(hits)   # :105-109
    return hr, dcg / idcg, ap


def epoch_schedule(epoch_count: int, start_epoch: int, start_test: int, test_freq: int,
                  start_store: Optional[int] = None, store_freq: Optional[int] = None) -> List[Tuple[int, bool, bool]]:
    """(epoch, should_test, should_store) per epoch (Helpers/ProcessController.py:42-75)."""
    end = start_epoch + epoch_count
    rows = []
    for cur in range(start_epoch, end):
        nxt = cur + 1
        t = (nxt - start_epoch >= start_test) and ((cur - start_test) % test_freq == 0 or nxt == end)
        s = False
        if start_store is not None and store_freq is not None:
            s = (nxt - start_epoch >= start_store) and ((cur - start_store) % store_freq == 0 or nxt == end)
        rows.append((cur, bool(t), bool(s)))
    return rows
