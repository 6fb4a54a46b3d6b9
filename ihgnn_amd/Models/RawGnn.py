"""The model (reference ``Models/RawGnn.py``): embeddings -> L hypergraph layers -> concat -> row gathers -> HEM.

Same constructor keywords, ``forward(user_indices, query_indices, item_indices=None)``,
``save_features_for_test`` / ``clear_saved_feature`` and state-dict keys (``embeddings.*``, ``gnn_{l}.*``,
``prediction_layer.items_bias``) as the reference, so its driver and checkpoints carry over.
"""
from typing import Optional, Type

import torch
import torch.nn as nn
from torch import Tensor

from ..Helpers.GlobalSettings import Gs
from .EmbeddingLayers import EmbeddingLayer
from .GnnLayers import GATLayer, GCNLayer, HGCNLayer, IHGNNLayer
from .PredictionLayers import HemPredictionLayer


class RawGnn(nn.Module):
    _saved_output_feature: Optional[Tensor] = None

    def __init__(self, device: torch.device, dataset, embedding_size: int, gnn_layer_type: Type,
                 gnn_layer_count: int, feature_interaction_order: int, phase2_attention: bool,
                 predictions: Type, lambda_muq: float):
        super().__init__()
        self.device = device
        self.dataset = dataset
        self.embedding_size = embedding_size
        self.gnn_layer_type = gnn_layer_type
        self.gnn_layer_count = gnn_layer_count
        self.feature_interaction_order = feature_interaction_order
        self.phase2_attention = phase2_attention
        self.prediction_layer_type = predictions
        self.output_feature_size = embedding_size * (1 + gnn_layer_count)
        # the width the kernels run at: the embedding size itself where it is one of the tiled widths (32 / 64 / 128 / 256), otherwise the next of them with zero columns
        # appended to the features and zero rows / columns to the weights (ops.padded_width: 96 -> 128, 160 / 192 / 224 -> 256; the reference takes any --emb,
        # Helpers/ArgsParser.py:94-95) - parameters and state-dict keep the embedding size
        from .. import ops as _ops
        self.compute_width = _ops.padded_width(embedding_size)
        if self.compute_width * (1 + gnn_layer_count) > self.MAX_SCORED_WIDTH:
            self.compute_width = embedding_size                  # (padding would push the scored row past the evaluation kernel's limit: the any-width kernels instead)
        if self.output_feature_size > self.MAX_SCORED_WIDTH:
            # refused HERE, not at the first evaluation after an epoch of training: the scoring kernel (ihg_score_topk) keeps the mixed rows of 32
            # (user, query) pairs in LDS for its whole run, which bounds the feature width d (L + 1)
            raise NotImplementedError(f'RawGnn: feature width embedding_size * (layers + 1) = {self.output_feature_size} exceeds {self.MAX_SCORED_WIDTH}, the widest '
                                      'matrix the MI355X evaluation kernel scores (32 mixed rows in 160 KB of LDS); use a smaller embedding or fewer layers')

        self.embeddings = EmbeddingLayer(dataset=dataset, embedding_size=embedding_size)

        self.gnns = []
        for depth in range(gnn_layer_count):
            if gnn_layer_type is IHGNNLayer:
                # only the first layer interacts at the requested order; deeper layers are first-order (RawGnn.py:76-78)
                order = feature_interaction_order if depth == 0 else min(feature_interaction_order, 1)
                layer = IHGNNLayer(device=device, dataset=dataset, input_dimension=embedding_size,
                                   output_dimension=embedding_size, feature_interaction_order=order,
                                   phase2_attention=phase2_attention)
            elif gnn_layer_type in (HGCNLayer, GCNLayer, GATLayer):
                layer = gnn_layer_type(device=device, dataset=dataset, input_dimension=embedding_size,
                                       output_dimension=embedding_size)
            else:
                raise NotImplementedError(f'unsupported GNN layer type: {gnn_layer_type}')
            self.gnns.append(layer)
            self.add_module(f'gnn_{depth}', layer)

        if predictions is not HemPredictionLayer:
            raise NotImplementedError(f'unsupported prediction layer type: {predictions}')
        self.prediction_layer = HemPredictionLayer(feature_dimension=self.output_feature_size, lambda_muq=lambda_muq,
                                                   item_count=dataset.item_count)

    # bce_loss: evaluate the last layer's hyperedge -> node pass only at the rows the loss reads (same loss, same gradients)
    batch_rows_only_last_layer = True
    MAX_SCORED_WIDTH = 1264                                   # = ihg_score_topk_max_dim() (tests/test_abi.py): 32 mixed rows as two fp16 planes in 160 KB of LDS

    def _compact_layout(self):
        """The hypergraph layout when it numbers its nodes WITHOUT the isolated ones (``IncidenceLayout.compact``: config C5, where two thirds of the nodes are in no
        hyperedge), else ``None``.  Every layer output of an isolated node is exactly zero (an empty sum times ``Dv^-1``), so the layers run on the
        layout's N' nodes and this class translates at its edges: X0 is gathered from the public ``[N, d]`` rows, the batch tail reads layer 0 by public row and the layers
        above through ``layout.node_map``, public callers get ``[N, d]`` matrices with zero rows put back."""
        lay = getattr(self.gnns[0], 'layout', None) if self.gnns else None
        return lay if (lay is not None and getattr(lay, 'compact', False)) else None

    def propagate_layers(self, tail_gradients=None, batch_rows=None, restrict_last_layer=True):
        """Full-graph propagation: the list ``[X0, X1, ..., XL]`` of ``[N, d]`` node features (input embeddings and every
        layer's output).  With ``tail_gradients`` (an ``ops.TailGradients``) every output is tapped: the returned tensors
        are the batch tail's halves, whose gradients travel through the holder instead of dense ``[N, d]`` tensors.
        ``batch_rows`` (int64 node rows): nobody reads ``XL`` outside these rows (a training step scores the batch only).  With
        ``restrict_last_layer`` the last hypergraph layer computes just them (plus the split rows of its plan) and leaves the rest
        unwritten; without it every row is computed and the layer is only told that its cotangent is zero outside them.

        Over a layout without the isolated nodes (``_compact_layout``): with ``tail_gradients`` the halves above layer 0 are ``[N', d]`` in the LAYOUT's numbering (the
        holder carries the row map; ``bce_loss`` is their only consumer), without it every output is returned in the public numbering."""
        from .. import ops
        last = len(self.gnns)
        x = None
        padded = self.compute_width != self.embedding_size
        compact = self._compact_layout()
        if (tail_gradients is not None and last >= 1 and ops.NODE_TABLES and not padded and compact is None
                and (batch_rows is None or int(batch_rows.shape[0]) <= ops.SCATTER_CHUNK_ROWS)):
            # a training step through the fused batch tail: X0 is not assembled - the first layer's transform and the tail read the embedding tables in place
            x = self.embeddings.node_tables(tail_gradients)
        tables = None
        if (tail_gradients is not None and last >= 1 and ops.NODE_TABLES and not padded and compact is not None
                and (batch_rows is None or int(batch_rows.shape[0]) <= ops.SCATTER_CHUNK_ROWS)):
            # ... over a layout without the isolated nodes: the active nodes' rows are gathered straight from the tables ([N', d]; the public [N, d] is never assembled), the
            # tail reads its layer-0 rows from the tables in place and the gather's backward adds their gradients to the tables' gradients
            tables = self.embeddings.node_tables(tail_gradients)
            if tables is not None:
                x = ops.gather_active_nodes(tables, compact)
        if x is None:
            x = self.embeddings.all_nodes()
            if padded:
                x = ops.pad_columns(x, self.compute_width)   # zero columns up to the next tiled width; every layer keeps them zero (its weights are padded with zeros)
        layer_rows = batch_rows
        if compact is not None and batch_rows is not None:
            # the rows the LAYERS are told (computed / differentiated rows of the last layer): the layout's own; an isolated batch node maps to row 0 - one more row listed
            layer_rows = compact.compact_rows(batch_rows, isolated_to=0)
        outputs = []
        for depth in range(last + 1):
            sparse_top = None
            if depth > 0:
                layer = self.gnns[depth - 1]
                if depth == last and batch_rows is not None and isinstance(layer, (IHGNNLayer, HGCNLayer)):
                    rows = getattr(layer_rows, 'as_int32', None)
                    if rows is None:
                        rows = layer_rows.to(torch.int32)
                    x = layer(x, output_rows=rows) if restrict_last_layer else layer(x, cotangent_rows=rows)
                    # the layer's backward pulls the listed rows of its cotangent only (the masked two-hop pull): the tap writes them and fills nothing
                    if ((restrict_last_layer or ops.SPARSE_LAST_COTANGENT) and not ops.CHECK_SPARSE_COTANGENT and layer.reads_cotangent_rows_only()
                            and int(batch_rows.shape[0]) <= ops.SCATTER_CHUNK_ROWS):
                        sparse_top = layer.layout
                else:
                    x = layer(x)
            if tail_gradients is not None:
                if isinstance(x, ops.NodeTables):
                    outputs.append(x)                        # (no tap: the first layer's transform adds the tail's layer-0 gradients itself)
                    continue
                if depth == 0 and tables is not None:
                    outputs.append(tables)                   # (no tap either: the gather's backward adds them; x is already the layout's [N', d])
                    continue
                x, for_tail = ops.tap(x, tail_gradients, depth, sparse_top)
                outputs.append(for_tail)
            elif compact is not None and depth > 0:
                outputs.append(self._to_public_rows(x, compact))
            else:
                outputs.append(x)
            if compact is not None and depth == 0 and tables is None:
                x = x.index_select(0, compact.active_nodes)  # X0 of the nodes that have hyperedges: what the layers run on
        return outputs

    @staticmethod
    def _to_public_rows(x: Tensor, compact) -> Tensor:
        """``[N', d]`` in the layout's numbering -> ``[N, d]`` in the public one, zero rows for the isolated nodes (differentiable)."""
        return torch.zeros(compact.public_node_count, x.shape[1], dtype=x.dtype, device=x.device).index_copy(0, compact.active_nodes, x)

    def propagate(self) -> Tensor:
        """``[N, d*(L+1)]``: all layer outputs side by side (``RawGnn.py:122``).  Outside autograd (the evaluation cache,
        ``save_features_for_test``) the matrix is allocated once and every layer WRITES ITS COLUMN SLICE of it - the kernels take row
        strides, layer l reads columns ``(l-1) d ..`` and writes columns ``l d ..`` of the same buffer - so there is no concatenation
        pass over ``[N, D]``.  Under autograd the outputs are separate tensors and are concatenated."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._without_padding_columns(torch.cat(self.propagate_layers(), 1))
        d, cw = self.embedding_size, self.compute_width
        w = self.embeddings.embedding_bag_vocabulary.weight
        features = torch.empty(self.dataset.node_count, cw * (1 + self.gnn_layer_count), dtype=torch.float32, device=w.device)
        self.embeddings.all_nodes(out=features[:, :d])
        if cw != d:
            features[:, d:cw].zero_()                        # (a width between the tiled ones: zero padding columns, kept zero by every layer; the scores do not see them)
        x = features[:, :cw]
        compact = self._compact_layout()
        if compact is not None:
            # the layers run on the layout's N' nodes (the ones that have hyperedges); their rows go back to the public matrix, the isolated nodes' rows are zero
            x = x.index_select(0, compact.active_nodes)
            features[:, cw:].zero_()
            for depth, layer in enumerate(self.gnns, 1):
                x = layer(x)
                features[:, depth * cw:(depth + 1) * cw].index_copy_(0, compact.active_nodes, x)
            return self._without_padding_columns(features)
        for depth, layer in enumerate(self.gnns, 1):
            x = layer(x, out=features[:, depth * cw:(depth + 1) * cw])
        return self._without_padding_columns(features)

    def _without_padding_columns(self, features: Tensor) -> Tensor:
        """``[N, cw (L + 1)]`` computed at a padded width -> the reference's ``[N, d (L + 1)]`` (``RawGnn.py:122``): the zero columns between the layers' blocks cut out."""
        d, cw = self.embedding_size, self.compute_width
        if cw == d:
            return features
        n = features.shape[0]
        return features.view(n, 1 + self.gnn_layer_count, cw)[:, :, :d].reshape(n, self.output_feature_size)

    def forward(self, user_indices: Tensor, query_indices: Tensor, item_indices: Optional[Tensor] = None) -> Tensor:
        """Indices are 0-based per type.  ``item_indices=None`` scores the given (user, query) against every item."""
        ds = self.dataset
        if self._saved_output_feature is None and item_indices is not None:
            # Training step.  The reference concatenates all layer outputs into [N, D] and gathers three [B, D] row sets
            # from it (RawGnn.py:122-131); here the 3B rows are gathered from each layer's [N, d] output with ONE index
            # op and concatenated afterwards ([3B, D] instead of [N, D]): same values, and the backward is one
            # scatter per layer instead of three dense [N, D] gradients that autograd would have to add up.
            rows = torch.cat([user_indices, query_indices + ds.query_start_index_in_graph,
                              item_indices + ds.item_start_index_in_graph])
            layers = self.propagate_layers()
            head = self.prediction_layer
            if layers[0].is_cuda and not Gs.Prediction.use_cosine_similarity and len(layers) <= 8:
                from .. import ops
                return ops.hem_score(layers, rows, item_indices, head.items_bias, head.lambda_muq, ds.item_start_index_in_graph)      # fused batch tail
            picked = torch.cat([x[rows] for x in layers], 1)
            b = user_indices.shape[0]
            return head(picked[:b], picked[b:2 * b], picked[2 * b:], item_indices)
        features = self._saved_output_feature if self._saved_output_feature is not None else self.propagate()
        if item_indices is None:
            item_feature = features[ds.item_start_index_in_graph:]
            if user_indices.dim() == 1 and user_indices.numel() > 1 and user_indices.stride(0) == 0 and query_indices.stride(0) == 0:
                user_indices, query_indices = user_indices[:1], query_indices[:1]      # one pair broadcast over all items
        else:
            item_feature = features[item_indices + ds.item_start_index_in_graph]
        user_feature = features[user_indices]
        query_feature = features[query_indices + ds.query_start_index_in_graph]
        return self.prediction_layer(user_feature, query_feature, item_feature, item_indices)

    def bce_loss(self, user_indices: Tensor, query_indices: Tensor, item_indices: Tensor, labels: Tensor, cotangent_sync=None) -> Tensor:
        """``nn.BCEWithLogitsLoss()(self(u, q, i), labels)`` with scoring, loss and their backward fused into the batch-tail
        kernels (the training loops use it when the loss function is a plain ``BCEWithLogitsLoss``).

        ``cotangent_sync`` (an ``ihgnn_amd.distributed.CotangentSync``; data parallel): this rank's loss on ITS batch, with a backward that yields the gradient of the
        MEAN of all ranks' losses - the ranks exchange the batch rows' cotangents (not the dense gradients) and every rank runs the propagation backward on their
        union; the layers are told the union of the ranks' batch rows (where the last layer's output is read / its cotangent is non-zero).  A COLLECTIVE: every
        rank calls it, with batches of the same size."""
        from .. import ops
        ds, head = self.dataset, self.prediction_layer
        rows = ops.batch_node_rows(user_indices, query_indices, item_indices, ds.query_start_index_in_graph, ds.item_start_index_in_graph)
        if cotangent_sync is not None and torch.is_grad_enabled():
            holder = ops.TailGradients(exchange=cotangent_sync.exchange, grad_scale=1.0 / cotangent_sync.world_size)
            read_rows = cotangent_sync.gather_rows(rows)           # the union of the ranks' batch rows (users | queries | items), int64 with .as_int32
        else:
            holder = ops.TailGradients() if torch.is_grad_enabled() else None
            read_rows = rows
        compact = self._compact_layout()
        if compact is None:
            return ops.hem_bce_loss(self.propagate_layers(holder, read_rows, self.batch_rows_only_last_layer), rows, item_indices, labels,
                                    head.items_bias, head.lambda_muq, ds.item_start_index_in_graph, holder)
        if holder is None:
            # (no gradient wanted: the public matrices, the plain tail)
            return nn.functional.binary_cross_entropy_with_logits(
                ops.hem_score(self.propagate_layers(), rows, item_indices, head.items_bias, head.lambda_muq, ds.item_start_index_in_graph), labels.float())
        holder.row_map = compact.node_map
        return ops.hem_bce_loss(self.propagate_layers(holder, read_rows, self.batch_rows_only_last_layer), rows, item_indices, labels,
                                head.items_bias, head.lambda_muq, ds.item_start_index_in_graph, holder, rows_upper=compact.compact_rows(rows))

    def supports_fused_loss(self, loss_function) -> bool:
        return (isinstance(loss_function, nn.BCEWithLogitsLoss) and loss_function.reduction == 'mean' and loss_function.weight is None
                and loss_function.pos_weight is None and self._saved_output_feature is None and not Gs.Prediction.use_cosine_similarity
                and len(self.gnns) + 1 <= 8 and next(self.parameters()).is_cuda)

    def score_all_items(self, user_indices: Tensor, query_indices: Tensor) -> Tensor:
        """Scores of ``C`` (user, query) pairs against every item as a dense ``[C, I]`` matrix (torch; a checker for tests and
        notebooks - the evaluation loop uses ``top_items``, which never builds that matrix).

        Same arithmetic as ``forward(u * ones(I), q * ones(I), None)`` per pair (``RawGnn.py:124-137`` +
        ``PredictionLayers.py:35-43``), batched: ``(lam*F[q] + (1-lam)*F[u]) @ F_items^T + bias``."""
        features = self._saved_output_feature if self._saved_output_feature is not None else self.propagate()
        ds, head = self.dataset, self.prediction_layer
        item_feature = features[ds.item_start_index_in_graph:]
        lam = head.lambda_muq
        mixed = lam * features[query_indices + ds.query_start_index_in_graph] + (1 - lam) * features[user_indices]
        return torch.addmm(head.items_bias.unsqueeze(0), mixed, item_feature.t())

    def top_items(self, user_indices: Tensor, query_indices: Tensor, k: int = 10):
        """``(items [C, k] int32, scores [C, k])``: the ``k`` best items of each (user, query) pair over the whole catalogue, best
        first - what ``Metrics.calculate_on_all_items`` keeps of ``forward(u, q, None)`` (``Metrics.py:60-61``) - from the fused
        HIP scoring + running top-k kernel; the ``[C, I]`` scores are never stored.  Ties: ascending item id.  No torch path: any feature width
        ``d (L + 1)`` up to ``MAX_SCORED_WIDTH`` (checked at construction)."""
        from .. import ops
        features = self._saved_output_feature if self._saved_output_feature is not None else self.propagate()
        ds, head = self.dataset, self.prediction_layer
        k = min(k, ds.item_count)
        if k > 10:
            raise NotImplementedError('RawGnn.top_items keeps at most ten items per pair (the reference reports HR / NDCG / MAP @10, Metrics.py:60-88)')
        return ops.score_topk(features, user_indices, query_indices, ds.query_start_index_in_graph, ds.item_start_index_in_graph,
                              head.items_bias, head.lambda_muq, k)

    def save_features_for_test(self) -> None:
        """Cache one propagation for the evaluation loop (call under ``torch.no_grad()``)."""
        self._saved_output_feature = self.propagate()

    def clear_saved_feature(self) -> None:
        self._saved_output_feature = None
