"""HEM scoring head (reference ``Models/PredictionLayers.py:6-44``): ``score = item . (lam*query + (1-lam)*user) + bias``."""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.init as init
from torch import Tensor
from torch.nn.parameter import Parameter

from ..Helpers.GlobalSettings import Gs


class HemPredictionLayer(nn.Module):
    def __init__(self, feature_dimension: int, lambda_muq: float, item_count: int):
        super().__init__()
        self.feature_dimension = feature_dimension
        self.lambda_muq = lambda_muq
        self.items_bias = Parameter(torch.empty(item_count))
        init.normal_(self.items_bias)

    def forward(self, user_feature: Optional[Tensor], query_feature: Tensor, item_feature: Tensor,
                item_indices: Optional[Tensor] = None) -> Tensor:
        bias = self.items_bias if item_indices is None else self.items_bias[item_indices]
        lam = self.lambda_muq
        mixed = query_feature if user_feature is None else lam * query_feature + (1 - lam) * user_feature
        if Gs.Prediction.use_cosine_similarity:
            return torch.cosine_similarity(item_feature, mixed) + bias
        if mixed.shape[0] == 1 and item_feature.shape[0] != 1:
            return torch.mv(item_feature, mixed[0]) + bias      # one (user, query) against many items
        return (item_feature * mixed).sum(1) + bias
