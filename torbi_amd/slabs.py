"""Reusable staging buffers of the many-file job.

The reference's loop allocates per batch: `pad_sequence` builds a new padded host tensor, `.to(device)` a new device
tensor (torbi/data/collate.py:24-31, torbi/core.py:193).  With PyTorch's caching allocators that is cheap only while
the sizes repeat; the batches of a ragged job all differ in size, several of them are in flight at once (one being read,
one being copied, two launch groups being decoded), and every miss is a `hipHostMalloc` / `hipMalloc` of 1.5-2.7 GB -- 8 ms
and more on the calling thread, a third of the job's run time before this pool existed (profiles/r03_file_job_profile.txt).

A `SlabPool` hands out uint8 buffers of at least the requested size and takes them back together with the event after
which they may be reused; a returned slab that is large enough serves any later request.  Pools live for the process
(`release()` empties them): the second job of a process allocates nothing.
"""
import threading
import weakref
from typing import List, Optional, Tuple

import torch


class SlabPool:
    """uint8 slabs in pinned host memory (`device=None`) or on a HIP device."""

    def __init__(self, device: Optional[torch.device] = None):
        self.device = device
        self._lock = threading.RLock()
        self._free: List[torch.Tensor] = []
        self._busy: List[Tuple[torch.Tensor, Optional[torch.cuda.Event]]] = []
        self._out = 0            # slabs handed out and not given back yet
        self._lent = {}          # id(slab) -> finalizer: a slab that dies without being given back stops counting as out

    def _collect(self) -> None:
        still = []
        for slab, event in self._busy:
            if event is None or event.query():
                self._free.append(slab)
            else:
                still.append((slab, event))
        self._busy = still

    def take(self, nbytes: int, limit: Optional[int] = None, exact: bool = False) -> torch.Tensor:
        """A slab of at least `nbytes` bytes: the smallest free one that fits, else a new allocation (sized generously,
        so that the slightly larger batch that follows fits as well).  With `limit`, a pool that already holds that
        many slabs waits for one in use to come back instead of growing (pinning a few GB costs more than the wait);
        a pool whose slabs are all too small replaces the smallest."""
        nbytes = max(int(nbytes), 1)
        while True:
            with self._lock:
                self._collect()
                fitting = [k for k, slab in enumerate(self._free) if slab.numel() >= nbytes]
                if fitting:
                    best = min(fitting, key=lambda k: self._free[k].numel())
                    return self._lend(self._free.pop(best))
                held = len(self._free) + len(self._busy) + self._out
                waiting_for = None
                if limit is not None and held >= limit:
                    pending = [event for slab, event in self._busy if event is not None and slab.numel() >= nbytes]
                    if pending:
                        waiting_for = pending[0]
                    elif self._free:
                        self._free.pop(min(range(len(self._free)), key=lambda k: self._free[k].numel()))
                    elif self._busy:
                        waiting_for = next((event for _, event in self._busy if event is not None), None)
            if waiting_for is None:
                break
            waiting_for.synchronize()
        # (`exact`: the caller asks for one size again and again -- the pinned allocator rounds what it is asked for up
        # to a power of two, so a generous 288 MB would pin 512)
        size = nbytes if exact else (nbytes + nbytes // 8 + (1 << 20) - 1) >> 20 << 20
        if self.device is None:
            slab = torch.empty((size,), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        else:
            slab = torch.empty((size,), dtype=torch.uint8, device=self.device)
        with self._lock:
            return self._lend(slab)

    def _lend(self, slab: torch.Tensor) -> torch.Tensor:
        """(lock held) Count `slab` as out until it is given back -- or dropped: a batch that takes a path which never
        returns its slabs (log_probs=False, the CPU route, an exception) must not make the pool believe for ever that it
        is at its limit and allocate a fresh multi-GB pinned buffer for every batch after."""
        self._out += 1
        key = id(slab)
        self._lent[key] = weakref.finalize(slab, self._lost, key)
        return slab

    def _lost(self, key: int) -> None:
        with self._lock:
            if self._lent.pop(key, None) is not None:
                self._out = max(0, self._out - 1)

    def give(self, slab: torch.Tensor, event: Optional[torch.cuda.Event] = None) -> None:
        """Return a slab; it is handed out again once `event` (recorded behind its last use) has completed."""
        with self._lock:
            finalizer = self._lent.pop(id(slab), None)
            if finalizer is not None:
                finalizer.detach()
                self._out = max(0, self._out - 1)
            self._busy.append((slab, event))

    def release(self) -> None:
        """Drop every slab that is not in use."""
        with self._lock:
            self._collect()
            self._free = []

    def held_bytes(self) -> int:
        with self._lock:
            return sum(s.numel() for s in self._free) + sum(s.numel() for s, _ in self._busy)


_pools = {}
_pools_lock = threading.RLock()


def pool(device: Optional[torch.device] = None) -> SlabPool:
    """The process-wide pool of pinned host slabs (`device=None`) or of slabs on `device`."""
    key = 'host' if device is None else str(torch.device(device))
    with _pools_lock:
        found = _pools.get(key)
        if found is None:
            found = _pools[key] = SlabPool(None if device is None else torch.device(device))
        return found


def release() -> None:
    """Free the staging buffers no job is using (they are kept between jobs otherwise)."""
    with _pools_lock:
        pools = list(_pools.values())
    for p in pools:
        p.release()
