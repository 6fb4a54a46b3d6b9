"""Hypergraph layers (reference ``Models/GnnLayers.py:118-236``): ``IHGNNLayer`` and ``HGCNLayer``.

Constructor signatures, sub-module names (``feature_transform``, ``feature_interactor.aggregation``) and the
absence of any non-linearity follow the reference; the sparse work runs in the HIP kernels of
libihgnn_hip.so through :mod:`ihgnn_amd.ops` instead of ``torch_sparse.matmul``.
``GCNLayer`` (pairwise-graph baseline) runs on the same segment-sum kernel over a weighted CSR; ``GATLayer`` (DGL) is
declared for the name tables only.
"""
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from .CommonLayers import FeatureInteractor


def _transform(linear: nn.Linear, x: Tensor, layout) -> Tensor:
    """``nn.Linear`` (square) on every node row through the HIP node-level kernels (split-arithmetic / MFMA row-GEMM for d in {32,64,128,256},
    the any-width kernels otherwise).  There is no torch path: a rectangular transform - which RawGnn never builds
    (``RawGnn.py:59-91``: input and output dimension are both the embedding size) - is refused."""
    if linear.in_features != linear.out_features:
        raise NotImplementedError(f'feature_transform {linear.in_features} -> {linear.out_features}: the MI355X path has square node-level transforms only '
                                  '(RawGnn builds every layer with input_dimension == output_dimension)')
    wide = _row_width(x)
    if wide != linear.in_features:
        # features zero-padded to the next tiled width (RawGnn at a width between 32 / 64 / 128 / 256, ops.padded_width): the weight as the top-left block of a zero matrix
        return ops.node_linear(x, ops.pad_square(linear.weight, wide), ops.pad_vector(linear.bias, wide), layout)
    return ops.node_linear(x, linear.weight, linear.bias, layout)


def _row_width(x) -> int:
    return int(x.shape[1])


def _check_rows(x, layout) -> None:
    """A layer's input has one row per node of ITS layout - which, for a hypergraph whose isolated nodes the layout leaves out (``IncidenceLayout.compact``), is not the
    public node count: ``RawGnn`` translates (``RawGnn._compact_layout``); a direct caller must pass ``x[layout.active_nodes]``."""
    if int(x.shape[0]) != layout.node_count:
        raise ValueError(f'layer input has {int(x.shape[0])} rows, the hypergraph layout {layout.node_count} nodes'
                         + (f' (the layout leaves out the {layout.public_node_count - layout.node_count} isolated nodes of the graph: pass x[layout.active_nodes], or build the '
                            'dataset under IHG_COMPACT_NODES=0)' if getattr(layout, 'compact', False) else ''))


class HGCNLayer(nn.Module):
    """``Y = Dv^-1/2 H De^-1 H^T Dv^-1/2 (X W^T + b)``  (``GnnLayers.py:142-153``)."""

    def __init__(self, device: torch.device, dataset, input_dimension: int, output_dimension: int):
        super().__init__()
        self.device = device
        self.dataset = dataset
        from ..layout import LogHyperLayout
        graph = dataset.graph if getattr(dataset, 'graph_type', None) is not None else dataset.hypergraph      # GnnLayers.py:131
        self.layout = graph.layout
        self.general = isinstance(self.layout, LogHyperLayout)  # per-search-log hyperedges of variable arity (PpsLogHyperGraph)
        if not self.general:
            self.edge_scale = float(torch.tensor(3.0).pow(-1))      # De^-1 of a 3-uniform hypergraph, as fp32
            self.out_scale = self.layout.inv_sqrt_deg * self.edge_scale
        self.feature_transform = nn.Linear(input_dimension, output_dimension)

    def reads_cotangent_rows_only(self) -> bool:
        """With ``output_rows`` / ``cotangent_rows`` the backward reads its cotangent at those rows only (``ops.node_two_hop``'s masked pull)."""
        return not self.general

    def forward(self, input_features: Tensor, output_rows: Optional[Tensor] = None, cotangent_rows: Optional[Tensor] = None,
                out: Optional[Tensor] = None) -> Tensor:
        """``out`` (inference only): the ``[N, d]`` destination, e.g. a column slice of the ``[N, d (L + 1)]`` feature matrix.
        ``output_rows`` (int32 node rows, not in the reference signature): the caller reads only these rows of the result
        (last layer of a training step); other rows may be left unwritten.  ``cotangent_rows``: every row is computed, but the caller
        promises that the gradient of the result is zero outside these rows (``ops.node_two_hop``)."""
        lay = self.layout
        _check_rows(input_features, lay)
        if self.general:
            # general incidence: node -> hyperedge (x Dv^-1/2 in, De^-1 out), hyperedge -> node (x Dv^-1/2 out): two K7 launches
            h = _transform(self.feature_transform, input_features, lay)
            edge_features = ops.hyper_node_to_edge(h, lay, src_scale=lay.inv_sqrt_deg, out_scale=lay.inv_edge_degree)
            return ops.hyper_edge_to_node(edge_features, lay, out_scale=lay.inv_sqrt_deg, out=out)
        h = _transform(self.feature_transform, input_features, lay)
        # node -> hyperedge -> node in one two-hop pass over the node table: Dv^-1/2 on the way in, Dv^-1/2 De^-1 on the way out
        return ops.node_two_hop(h, lay, in_scale=lay.inv_sqrt_deg, out_scale=self.out_scale, rows=output_rows, cotangent_rows=cotangent_rows, out=out)


class IHGNNLayer(nn.Module):
    """``Y = Dv^-1 H Interact(X W^T + b)``  (``GnnLayers.py:221-236``, phase-2 attention off)."""

    def __init__(self, device: torch.device, dataset, input_dimension: int, output_dimension: int,
                 feature_interaction_order: int, phase2_attention: bool):
        super().__init__()
        if feature_interaction_order not in (1, 2, 3):
            raise AssertionError('feature interaction order must be 1, 2 or 3')
        if phase2_attention:
            raise NotImplementedError('phase-2 attention (DGL edge-softmax branch, GnnLayers.py:200-216) is outside '
                                      'the MI355X hypergraph path; the reference driver hard-codes it off (Main.py:57)')
        self.device = device
        self.dataset = dataset
        self.feature_interaction_order = feature_interaction_order
        self.attention_phase2 = False
        self.layout = dataset.hypergraph.layout
        self.feature_interactor = FeatureInteractor(dataset=dataset, max_order=feature_interaction_order,
                                                    node_feature_dimension=input_dimension,
                                                    output_dimension=input_dimension)
        self.feature_transform = nn.Linear(input_dimension, output_dimension)

    def reads_cotangent_rows_only(self) -> bool:
        """First-order layers run the two-hop operator, whose backward under ``output_rows`` / ``cotangent_rows`` is the masked pull."""
        return self.feature_interaction_order == 1

    def forward(self, input_features: Tensor, output_rows: Optional[Tensor] = None, cotangent_rows: Optional[Tensor] = None,
                out: Optional[Tensor] = None) -> Tensor:
        """``output_rows`` / ``cotangent_rows`` / ``out``: as in ``HGCNLayer.forward``."""
        _check_rows(input_features, self.layout)
        if self.feature_interaction_order == 1:
            # first-order layer: hoisted node-level blocks, then node -> hyperedge -> node fused into one two-hop pass
            return ops.node_two_hop(self._first_order_of_input(input_features), self.layout, out_scale=self.layout.inv_deg, rows=output_rows,
                                    cotangent_rows=cotangent_rows, out=out)
        h = _transform(self.feature_transform, input_features, self.layout)
        return self.feature_interactor.to_nodes(h, out_scale=self.layout.inv_deg, rows=output_rows, out=out)

    def _first_order_of_input(self, x: Tensor) -> Tensor:
        """``first_order(feature_transform(x))``.  With no non-linearity between them (``GnnLayers.py:224-227`` +
        ``CommonLayers.py:60-66``) the two linear maps of a first-order layer compose: node type t sees
        ``x (A_t W)^T + (A_t b + [t = user] c)``, one typed row-GEMM over ``[N, d]`` instead of two (and one instead of two in
        each backward direction); the ``[d, d]`` products are formed per call by a tiny kernel (``ihg_compose_first_order_fwd``).  Same reassociation class as the hoisting itself:
        <= 5e-7 relative against the reference's order of operations."""
        lin, agg = self.feature_transform, self.feature_interactor.aggregation
        if lin.bias is None:
            return self.feature_interactor.first_order(_transform(lin, x, self.layout))
        wide = _row_width(x)                                 # (> in_features: zero-padded features, ops.padded_width)
        weight, bias = ops.compose_first_order(ops.pad_blocks(agg.weight, wide), ops.pad_vector(agg.bias, wide), ops.pad_square(lin.weight, wide),
                                               ops.pad_vector(lin.bias, wide))      # [d, 3 d]: A_t W side by side; [3, d]
        return ops.node_linear(x, weight, bias, self.layout, typed=True, bias_mask=0b111)


class GCNLayer(nn.Module):
    """``Y = D^-1/2 A D^-1/2 (X W^T + b)`` over the pairwise graph (``GnnLayers.py:9-45``).

    The reference orders transform and propagation by which side is narrower (``GnnLayers.py:33-43``); RawGnn builds square layers
    (transform first).  Propagation is the weighted-CSR form of the K7 kernel with both ``D^-1/2`` scalings fused in."""

    def __init__(self, device: torch.device, dataset, input_dimension: int, output_dimension: int):
        super().__init__()
        self.device = device
        self.dataset = dataset
        self.input_dimension = input_dimension
        self.output_dimension = output_dimension
        self.graph = dataset.graph2d.layout
        self.feature_transform = nn.Linear(input_dimension, output_dimension)

    def forward(self, input_features: Tensor, out: Optional[Tensor] = None) -> Tensor:
        # the reference transforms first when the output is not wider (GnnLayers.py:33-43); RawGnn's layers are square, so that is the only order here
        return ops.pair_spmm(_transform(self.feature_transform, input_features, self.graph), self.graph, out=out)


class GATLayer(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError('GATLayer (DGL edge-softmax baseline, GnnLayers.py:48-115) is outside the MI355X hypergraph '
                                  'path (SURVEY.md §2); use IHGNNLayer, HGCNLayer or GCNLayer')
