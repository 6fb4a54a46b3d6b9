"""The "interactive" node -> hyperedge step (reference ``Models/CommonLayers.py:29-87``).

``FeatureInteractor`` owns the reference's ``aggregation = nn.Linear(k*d, d)`` (k = 3 / 6 / 7 for interaction
order 1 / 2 / 3; column blocks u, q, i, uq, qi, iu, uqi) but never builds the ``[E, k*d]`` concatenation:

* the three first-order blocks are applied per NODE, before the gather (users only ever meet block u, queries
  block q, items block i): ``P = [H_users W_u^T + c ; H_queries W_q^T ; H_items W_i^T]`` - ``N*d*d`` flops
  instead of ``E*3*d*d`` - and the hyperedge value is the gather-sum ``P[u] + P[q] + P[i]`` (HIP kernel K5);
* the product blocks (orders 2, 3) are contracted inside the fused HIP kernel ``ihg_interact_fwd``.
"""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
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
        d = self.node_feature_dimension
        w, b = self.aggregation.weight, self.aggregation.bias
        if ops.node_linear_supported(node_features, w):
            return ops.node_linear(node_features, w, b, self.dataset.hypergraph.layout, typed=True, bias_mask=0b001)
        u_end = self.dataset.query_start_index_in_graph
        q_end = self.dataset.item_start_index_in_graph
        return torch.cat([F.linear(node_features[:u_end], w[:, :d], b),
                          F.linear(node_features[u_end:q_end], w[:, d:2 * d]),
                          F.linear(node_features[q_end:], w[:, 2 * d:3 * d])])

    def to_nodes(self, node_features: Tensor, out_scale: Optional[Tensor] = None, rows: Optional[Tensor] = None) -> Tensor:
        """``out_scale * H forward(node_features)``: the hyperedge features taken on to the nodes (``GnnLayers.py:229-236``).  Orders 2 / 3
        on the hoisted path run as one autograd node whose backward forms the hyperedges' cotangents inside the member-gradient kernel."""
        layout = self.dataset.hypergraph.layout
        if self.max_order > 1 and not ops.interact_from_nodes_supported(node_features, self.aggregation.weight):
            return ops.interact_to_nodes(node_features, self.first_order(node_features), self.aggregation.weight, layout, self.max_order, out_scale, rows)
        return ops.node_segment_sum(self(node_features), layout, out_scale=out_scale, rows=rows)

    def forward(self, node_features: Tensor) -> Tensor:
        layout = self.dataset.hypergraph.layout
        if self.max_order > 1 and ops.interact_from_nodes_supported(node_features, self.aggregation.weight):
            return ops.interact_from_nodes(node_features, self.aggregation.weight, self.aggregation.bias, layout, self.max_order)
        hoisted = self.first_order(node_features)
        if self.max_order == 1:
            return ops.edge_gather_sum(hoisted, layout)
        return ops.interact(node_features, hoisted, self.aggregation.weight, layout, self.max_order)
