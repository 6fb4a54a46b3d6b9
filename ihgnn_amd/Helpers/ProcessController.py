"""Epoch iterator with test / checkpoint predicates and an ETA (reference ``Helpers/ProcessController.py:4-111``)."""
import math
from typing import Iterator, List, Optional


class ProcessController:
    def __init__(self, epoch_count: int, start_epoch: int, start_test_epoch: int, test_frequency: int,
                 start_store_epoch: Optional[int] = None, store_frequency: Optional[int] = None):
        self.StartEpoch = start_epoch
        self.EpochCount = epoch_count
        self.EndEpoch = start_epoch + epoch_count            # exclusive
        self.CurrentEpoch = start_epoch - 1
        self._test_from, self._test_every = start_test_epoch, test_frequency
        storing = start_store_epoch is not None and store_frequency is not None
        self._store_from = start_store_epoch if storing else None
        self._store_every = store_frequency if storing else None
        self._planned_tests = 1 + (epoch_count - start_test_epoch) / test_frequency
        self._train_times: List[float] = []
        self._test_times: List[float] = []

    def __len__(self) -> int:
        return self.EpochCount

    def __iter__(self) -> Iterator[int]:
        self.CurrentEpoch = self.StartEpoch - 1
        return self

    def __next__(self) -> int:
        self.CurrentEpoch += 1
        if self.CurrentEpoch == self.EndEpoch:
            raise StopIteration()
        return self.CurrentEpoch

    def _due(self, first: int, every: int) -> bool:
        done = self.CurrentEpoch + 1 - self.StartEpoch       # epochs finished in this run, current included
        on_grid = (self.CurrentEpoch - first) % every == 0
        return done >= first and (on_grid or self.CurrentEpoch + 1 == self.EndEpoch)

    def ShouldTest(self) -> bool:
        return self._due(self._test_from, self._test_every)

    def ShouldStore(self) -> bool:
        return self._store_from is not None and self._due(self._store_from, self._store_every)

    def AddTrainTime(self, seconds: float) -> None:
        self._train_times.append(seconds)

    def AddTestTime(self, seconds: float) -> None:
        self._test_times.append(seconds)

    def GetRemainingTime(self) -> float:
        if not self._train_times:
            return float('nan')
        recent = lambda xs: sum(xs[-2:]) / len(xs[-2:])
        epoch_t = recent(self._train_times)
        test_t = recent(self._test_times) if self._test_times else 2 * epoch_t
        return epoch_t * (self.EndEpoch - self.CurrentEpoch) + test_t * (self._planned_tests - len(self._test_times))

    def GetRemainingTimeString(self) -> str:
        t = self.GetRemainingTime()
        if math.isnan(t):
            return 'n/a'
        if t >= 3600:
            return f'{int(t // 3600)} h {int(t / 60 - 60 * (t // 3600))} m'
        return f'{int(t / 60)} m' if t >= 60 else f'{int(t)} s'
