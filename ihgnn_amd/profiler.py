"""Per-kernel device timing with HIP events on the launch stream.

Disabled (zero overhead beyond one attribute test) unless ``start()`` was called.  While enabled, every
kernel launch made through :mod:`ihgnn_amd.ops` is bracketed by a pair of timing events recorded on the
stream the kernel is enqueued on (torch's current stream, which is the stream handed to the C ABI), so
``summary()`` reports true device durations of launches inside a larger timed region.  ``bench.py`` uses it
for the ``roofline`` object; the numbers agree with ``rocprofv3 --kernel-trace --stats`` (profiles/).
"""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager
from typing import Dict, List, Tuple

import torch

_enabled = False
_records: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event, int, int]]] = defaultdict(list)
_only = None


def start(only=None) -> None:
    """Begin recording; ``only`` restricts recording to a set of kernel names."""
    global _enabled, _only
    _records.clear()
    _only = set(only) if only else None
    _enabled = True


def stop() -> None:
    global _enabled
    _enabled = False


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _Null()


@contextmanager
def _timed(name: str, rows: int, dim: int):
    begin = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    begin.record()
    try:
        yield
    finally:
        end.record()
        _records[name].append((begin, end, rows, dim))


def kernel(name: str, rows: int, dim: int):
    if not _enabled or (_only is not None and name not in _only):
        return _NULL
    return _timed(name, rows, dim)


def summary() -> Dict[str, dict]:
    """``{kernel: {launches, total_ms, avg_us, rows, dim}}`` for everything recorded since ``start()``."""
    torch.cuda.synchronize()
    out = {}
    for name, recs in _records.items():
        times = [b.elapsed_time(e) for b, e, _, _ in recs]
        out[name] = dict(launches=len(recs), total_ms=sum(times), avg_us=1e3 * sum(times) / max(len(times), 1),
                         rows=recs[-1][2], dim=recs[-1][3], times_ms=times)
    return out
