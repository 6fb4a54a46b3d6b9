"""Input node features (reference ``Models/EmbeddingLayers.py``).

Same parameters and state-dict keys as the reference (``embedding_user.weight [U+1,d]``,
``embedding_item.weight [I+1,d]``, ``embedding_bag_vocabulary.weight [V+1,d]``; row 0 of each is the
padding row, xavier-uniform initialised), but the full-graph lookup that every training step performs
(``EmbeddingLayer(None, None, None)``, ``RawGnn.py:112``) is not three gather kernels:

* users / items: ids are exactly ``1..U`` / ``1..I`` (``Dataset.py:153-154``), so the "lookup" is the view
  ``weight[1:]`` - no kernel, and its backward is a plain dense gradient;
* queries: the HIP embedding-bag mean kernel over the CSR of query words (``ihg_bag_mean_fwd/bwd``).
"""
from typing import Optional, Tuple

import torch.nn as nn
import torch.nn.init as init
from torch import Tensor

from .. import ops
from ..Helpers.GlobalSettings import Gs, Gsv


class EmbeddingLayer(nn.Module):
    def __init__(self, dataset, embedding_size: int):
        super().__init__()
        if Gs.Query.transform != Gsv.mean:
            raise NotImplementedError('only the mean query transform is on the MI355X path (GlobalSettings.py:71-73)')
        self.dataset = dataset
        self.embedding_size = embedding_size
        self.embedding_user = EmbeddingLayer.create_embedding(dataset.user_count + 1, embedding_size, padding_idx=0)
        self.embedding_item = EmbeddingLayer.create_embedding(dataset.item_count + 1, embedding_size, padding_idx=0)
        self.embedding_bag_vocabulary = EmbeddingLayer.create_embedding_bag(dataset.vocab_size + 1, embedding_size)

    def forward(self, user_indices: Optional[Tensor] = None, query_indices: Optional[Tensor] = None,
                item_indices: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
        return self.embed_user(user_indices), self.embed_query(query_indices), self.embed_item(item_indices)

    def all_nodes(self, out: Optional[Tensor] = None) -> Tensor:
        """``torch.cat(self(None, None, None))`` as one op (``RawGnn.py:112-113``): ``[U+Q+I, d]``.  ``out`` (inference only): write into
        this ``[N, d]`` destination (a column slice of the feature matrix)."""
        return ops.embed_all_nodes(self.embedding_user.weight, self.embedding_item.weight, self.embedding_bag_vocabulary.weight, self.dataset.bag_layout, out)

    def node_tables(self, holder):
        """The same features WITHOUT assembling them (``ops.NodeTables``): the first layer's node-level transform reads the tables in place and its backward writes
        their gradients in place; ``None`` where the library does not offer that (other widths, fp32-MFMA arithmetic)."""
        tables = (self.embedding_user.weight, self.embedding_item.weight, self.embedding_bag_vocabulary.weight)
        if not ops.NodeTables.supported(*tables):
            return None
        return ops.NodeTables(*tables, self.dataset.bag_layout, holder)

    def embed_user(self, user_indices: Optional[Tensor] = None) -> Tensor:
        w = self.embedding_user.weight
        return w[1:] if user_indices is None else w[user_indices + 1]

    def embed_item(self, item_indices: Optional[Tensor] = None) -> Tensor:
        w = self.embedding_item.weight
        return w[1:] if item_indices is None else w[item_indices + 1]

    def embed_query(self, query_indices: Optional[Tensor] = None) -> Tensor:
        queries = ops.bag_mean(self.embedding_bag_vocabulary.weight, self.dataset.bag_layout)
        return queries if query_indices is None else queries[query_indices]

    @staticmethod
    def create_embedding(num_embeddings: int, embedding_dimension: int, padding_idx: Optional[int] = None) -> nn.Embedding:
        table = nn.Embedding(num_embeddings, embedding_dimension, padding_idx=padding_idx)
        init.xavier_uniform_(table.weight)
        return table

    @staticmethod
    def create_embedding_bag(num_embeddings: int, embedding_dimension: int, mode: str = 'mean') -> nn.EmbeddingBag:
        table = nn.EmbeddingBag(num_embeddings, embedding_dimension, mode=mode)
        init.xavier_uniform_(table.weight)
        return table
