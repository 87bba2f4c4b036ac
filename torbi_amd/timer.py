"""The user-visible timing scope of the decode: the stand-in for `torchutil.time` as the reference uses it.

The reference wraps its operator call in `torchutil.time.context('torbi')` (torbi/core.py:200-206) and its evaluation
reads the totals back with `torchutil.time.results()` / clears them with `torchutil.time.reset()`
(torbi/evaluate/core.py:40,118).  Same three names here: `context(name)` accumulates the wall-clock time spent inside
the block under `name`, `results()` returns `{name: seconds}`, `reset()` clears.  `from_probabilities` opens
`context('torbi')` around its decode exactly where upstream does.

GPU work is asynchronous: like upstream's, the scope only covers what happened on the host unless the block synchronises.
`context(name, device=...)` brackets the block with HIP events on that device's current stream instead and adds the
DEVICE time between them (read without blocking when `results()` is called; events that have not completed yet are
waited for there) -- the number the reference's published CUDA figures would have needed.
"""
import contextlib
import threading
import time
from typing import Dict, Optional

import torch

_lock = threading.Lock()
_totals: Dict[str, float] = {}
_pending = []            # (name, start event, end event)


@contextlib.contextmanager
def context(name: str, device: Optional[torch.device] = None):
    """Accumulate the time spent in the block under `name` (host wall clock; device time with `device`)."""
    if device is not None and torch.cuda.is_available():
        stream = torch.cuda.current_stream(device)
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record(stream)
        try:
            yield
        finally:
            end.record(torch.cuda.current_stream(device))
            with _lock:
                _pending.append((name, begin, end))
        return
    start = time.perf_counter()
    try:
        yield
    finally:
        elapsed = time.perf_counter() - start
        with _lock:
            _totals[name] = _totals.get(name, 0.0) + elapsed


def results() -> Dict[str, float]:
    """`{name: seconds}` accumulated since the last `reset()`."""
    with _lock:
        pending, _pending[:] = list(_pending), []
    for name, begin, end in pending:
        end.synchronize()
        seconds = begin.elapsed_time(end) * 1e-3
        with _lock:
            _totals[name] = _totals.get(name, 0.0) + seconds
    with _lock:
        return dict(_totals)


def reset() -> None:
    with _lock:
        _totals.clear()
        _pending.clear()
