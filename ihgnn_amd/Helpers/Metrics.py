"""HR@10 / NDCG@10 / MAP@10 of one search scored over all items, and the per-epoch collection that picks the
best epoch by validation NDCG@10 (reference ``Helpers/Metrics.py:8-162``).

Top-10 selection uses ``torch.topk`` on the score vector's own device (one small D2H copy of ten indices per
search instead of a full sort + copy, reference ``Metrics.py:60-61``); ties inside the top ten may order
differently from the reference's unstable ``torch.sort``.
"""
import math
from io import UnsupportedOperation
from typing import Any, Callable, Iterable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

TOP_K = 10
_LOG2_OF = [0.0, 0.0] + [math.log(2, r) for r in range(2, TOP_K + 2)]      # log_r(2), r = rank + 2


class Metrics:
    title = 'HitRatio@10 NDCG@10 MAP@10'

    def __init__(self, hit_ratio: float = 0.0, ndcg: float = 0.0, mean_ap: float = 0.0):
        self.HitRatio_at10 = hit_ratio
        self.NDCG_at10 = ndcg
        self.MAP_at10 = mean_ap

    # accumulation ---------------------------------------------------------------------------
    def add_to_self(self, other: 'Metrics') -> None:
        self.HitRatio_at10 += other.HitRatio_at10
        self.NDCG_at10 += other.NDCG_at10
        self.MAP_at10 += other.MAP_at10

    def divide_and_get_new(self, count) -> 'Metrics':
        return Metrics(self.HitRatio_at10 / count, self.NDCG_at10 / count, self.MAP_at10 / count)

    # formatting -----------------------------------------------------------------------------
    def to_string(self, highlight: bool = False, no_title: bool = False) -> str:
        if no_title:
            return f'{self.HitRatio_at10:.4f} {self.NDCG_at10:.4f} {self.MAP_at10:.4f}'
        row = f'{self.HitRatio_at10:<11.4f} {self.NDCG_at10:<7.4f} {self.MAP_at10:<6.4f}'
        if highlight:
            row = f'\033[0;41m{row}\033[0m'
        return self.title + '\n' + row

    def to_highlight_string(self) -> str:
        return self.to_string(highlight=True)

    __str__ = __repr__ = lambda self: self.to_string()

    # the metric itself ----------------------------------------------------------------------
    @staticmethod
    def from_top_indices(top: Sequence[int], interacted_items: Sequence[int], flags: Optional[Sequence[int]],
                         flags_are_all_1: bool) -> 'Metrics':
        """Metrics from the ten best item indices (best first)."""
        rank_of = {item: rank for rank, item in enumerate(top)}
        cap = min(len(interacted_items), TOP_K)
        if flags_are_all_1:
            hits = [rank_of[item] for item in interacted_items if item in rank_of]
            dcg = sum(_LOG2_OF[r + 2] for r in hits)
            idcg = Metrics._get_idcg_for_all1(cap)
        else:
            pairs = [(rank_of[item], f) for item, f in zip(interacted_items, flags) if item in rank_of]
            hits = [r for r, _ in pairs]
            dcg = Metrics._get_dcg(hits, [f for _, f in pairs])
            idcg = Metrics._get_idcg(sorted((f for _, f in pairs), reverse=True))
        return Metrics(len(hits) / cap, dcg / idcg, Metrics._get_map_for_all1(hits))

    @staticmethod
    def calculate_on_all_items(model_outputs: Tensor, interacted_items: List[int], flags: List[int],
                               flags_are_all_1: bool) -> 'Metrics':
        k = min(TOP_K, model_outputs.shape[0])
        top = torch.topk(model_outputs, k, largest=True, sorted=True).indices.tolist()
        return Metrics.from_top_indices(top, interacted_items, flags, flags_are_all_1)

    @staticmethod
    def _get_dcg(indices_hit, flags_hit) -> float:
        return sum(math.log(2, i + 2) * (2 ** r - 1) for i, r in zip(indices_hit, flags_hit))

    @staticmethod
    def _get_dcg_for_all1(indices_hit) -> float:
        return sum(math.log(2, i + 2) for i in indices_hit)

    @staticmethod
    def _get_idcg(flags_descending) -> float:
        return sum(math.log(2, i + 2) * (2 ** r - 1) for i, r in enumerate(flags_descending))

    @staticmethod
    def _get_idcg_for_all1(truth_count: int) -> float:
        return sum(math.log(2, r) for r in range(2, 2 + truth_count))

    @staticmethod
    def _get_map_for_all1(indices_hit) -> float:
        if not indices_hit:
            return 0
        return sum(j / (i + 1) for j, i in enumerate(indices_hit, start=1)) / len(indices_hit)


class MetricsCollection:
    """(epoch, test metrics[, valid metrics]) history."""

    def __init__(self, has_valid: bool = False):
        self._has_valid = has_valid
        self._epochs: List[int] = []
        self._tests: List[Metrics] = []
        self._valids: List[Metrics] = []

    @property
    def has_valid(self) -> bool:
        return self._has_valid

    def add(self, epoch: int, m_test: Metrics, m_valid: Optional[Metrics] = None) -> None:
        if self._has_valid != (m_valid is not None):
            raise ValueError(f'has_valid is {self._has_valid}.')
        if m_valid is not None:
            self._valids.append(m_valid)
        self._epochs.append(epoch)
        self._tests.append(m_test)

    def _best(self, pool: List[Metrics], key: Callable[[Metrics], Any], max_is_best: bool) -> int:
        scores = [key(m) for m in pool]
        return scores.index(max(scores) if max_is_best else min(scores))    # first best, like list.index(max(...))

    def get_valid_best(self, key, max_is_best: bool = True) -> Tuple[int, Metrics, Metrics]:
        if not self._has_valid:
            raise UnsupportedOperation('has_valid is False.')
        i = self._best(self._valids, key, max_is_best)
        return self._epochs[i], self._tests[i], self._valids[i]

    def get_test_best(self, key, max_is_best: bool = True):
        i = self._best(self._tests, key, max_is_best)
        if self._has_valid:
            return self._epochs[i], self._tests[i], self._valids[i]
        return self._epochs[i], self._tests[i]

    def iter_epoch_test_valid(self) -> Iterable[Tuple[int, Metrics, Metrics]]:
        if not self._has_valid:
            raise UnsupportedOperation('has_valid is False.')
        return zip(self._epochs, self._tests, self._valids)

    def iter_epoch_test(self) -> Iterable[Tuple[int, Metrics]]:
        return zip(self._epochs, self._tests)
