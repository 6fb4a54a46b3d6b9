"""The "interactive" node -> hyperedge step (reference ``Models/CommonLayers.py:29-87``).

``FeatureInteractor`` owns the reference's ``aggregation = nn.Linear(k*d, d)`` (k = 3 / 6 / 7 for interaction
order 1 / 2 / 3; column blocks u, q, i, uq, qi, iu, uqi) but never builds the ``[E, k*d]`` concatenation:

* the three first-order blocks are applied per NODE, before the gather (users only ever meet block u, queries
  block q, items block i): ``P = [H_users W_u^T + c ; H_queries W_q^T ; H_items W_i^T]`` - ``N*d*d`` flops
  instead of ``E*3*d*d`` - and the hyperedge value is the gather-sum ``P[u] + P[q] + P[i]`` (HIP kernel K5);
* the product blocks (orders 2, 3) are contracted inside the fused HIP kernel ``ihg_interact_fwd``.
"""
from typing import Optional

import torch.nn as nn
from torch import Tensor

from .. import ops


class FeatureInteractor(nn.Module):
    def __init__(self, dataset, max_order: int, node_feature_dimension: int, output_dimension: int):
        super().__init__()
        if max_order not in (1, 2, 3):
            raise AssertionError('feature interaction order must be 1, 2 or 3')
        self.dataset = dataset
        self.max_order = max_order
        self.node_feature_dimension = node_feature_dimension
        self.output_dimension = output_dimension
        blocks = {1: 3, 2: 6, 3: 7}[max_order]
        self.aggregation = nn.Linear(blocks * node_feature_dimension, output_dimension)

    def first_order(self, node_features: Tensor) -> Tensor:
        """Node-level image of the u / q / i blocks (+ bias, carried by the user rows: one user per hyperedge)."""
        w, b = self._operands(node_features)
        return ops.node_linear(node_features, w, b, self.dataset.hypergraph.layout, typed=True, bias_mask=0b001)      # any width; no torch path

    def _operands(self, node_features: Tensor):
        """``(aggregation.weight, aggregation.bias)`` at the width of ``node_features``: themselves, or - features zero-padded to the next tiled width, ``ops.padded_width`` -
        every ``[d, d]`` block in the top-left of a zero block of that width."""
        wide = int(node_features.shape[1])
        return ops.pad_blocks(self.aggregation.weight, wide), ops.pad_vector(self.aggregation.bias, wide)

    def to_nodes(self, node_features: Tensor, out_scale: Optional[Tensor] = None, rows: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
        """``out_scale * H forward(node_features)``: the hyperedge features taken on to the nodes (``GnnLayers.py:229-236``).  Orders 2 / 3
        run as ONE autograd node (``ops.interact_layer``): the backward forms the hyperedges' cotangents inside the member-gradient kernel where it
        can and adds the first-order path's input gradient onto the member gradients inside the node-level kernel."""
        layout = self.dataset.hypergraph.layout
        if self.max_order > 1:
            w, b = self._operands(node_features)
            return ops.interact_layer(node_features, w, b, layout, self.max_order, out_scale, rows, out)
        return ops.node_segment_sum(self(node_features), layout, out_scale=out_scale, rows=rows, out=out)

    def forward(self, node_features: Tensor) -> Tensor:
        """Hyperedge features ``[E', d]``, one row per hyperedge OF THE LAYOUT, in the layout's own numbering (by user): every interaction's row unless the layout keeps
        repeated (user, query, item) triples once (``layout.edge_weight``; ``layout.file_to_edge`` maps an interaction to its row) - the reference's ``[E, d]`` in file order
        is ``out[layout.file_to_edge]``.  ``node_features`` has one row per node of the layout (``layout.node_count``: without the isolated nodes where it is compact)."""
        layout = self.dataset.hypergraph.layout
        w, b = self._operands(node_features)
        if self.max_order > 1 and ops.interact_from_nodes_supported(node_features, w):
            return ops.interact_from_nodes(node_features, w, b, layout, self.max_order)
        hoisted = self.first_order(node_features)
        if self.max_order == 1:
            return ops.edge_gather_sum(hoisted, layout)
        return ops.interact(node_features, hoisted, w, layout, self.max_order)
