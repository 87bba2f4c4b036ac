"""Stream-level pipelining and grouping of consecutive decodes.

The reference's driver is fully serial (load -> decode -> save, torbi/core.py:417-457).  Batches are
independent, which buys two things here:

* **groups** (`group` > 1): consecutive batches are collected and decoded by ONE call of
  `torbi_amd.decode_batches` -- with enough items that is one time-resident forward launch for the whole
  group (csrc/resident_forward.hpp: 16 items per workgroup, posterior rows resident in the LDS, no per-timestep
  launches), 2.1x the per-timestep path on the 512 x 500 x 1440 benchmark with eight batches per group;
* **streams** (`depth` > 1): consecutive decodes (or groups) alternate between HIP streams with private
  scratch, so one's latency-bound backtrace and launch tails hide under the next one's forward pass
  (18.6 ms per 512x500x1440 decode against 20.3 ms on one stream with the per-timestep path).

`DecodePipeline` keeps `decode`'s contract per call; only completion is deferred: the returned indices are
valid once `wait(indices)` / `synchronize()` returns (or on the stream the pipeline used, for callers that
chain more GPU work).
"""
from typing import Callable, List, Optional

import torch

from . import viterbi


_streams = {}        # device -> side streams shared by every pipeline of the process


def _side_streams(device, depth):
    """The first `depth` side streams of `device`, created once per process.  HIP multiplexes streams onto a few
    hardware queues (4 by default); a second pipeline with streams of its own can land two of them on one queue,
    and launch groups that share a queue do not overlap (measured: the many-file job at 32 M instead of 49 M
    timesteps/s when it ran after another pipeline in the same process).  Streams only order work, so pipelines
    can share them."""
    pool = _streams.setdefault(str(device), [])
    while len(pool) < depth:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:depth]


class DecodePipeline:
    """Decodes in groups of `group` batches, round-robin over `depth` side streams with private scratch."""

    def __init__(self, device=None, depth: int = 2, reuse_preparation: bool = True, group: int = 1,
                 path: Optional[str] = None):
        if not torch.cuda.is_available():
            raise RuntimeError('DecodePipeline needs a HIP device; there is no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None \
            else torch.device(device)
        self.depth = max(1, int(depth))
        self.group = max(1, min(int(group), viterbi._lib.MAX_BATCHES))
        self.path = path
        # each slot's scratch is private, so the per-transition preparation of one decode can serve the next
        # decode of the same slot when shape and transition tensor are unchanged (decode(reuse_preparation=))
        self.reuse_preparation = bool(reuse_preparation)
        self.streams = _side_streams(self.device, self.depth)
        # scratch[slot][k]: workspace of the k-th batch of the group that runs on `slot`
        self.scratch: List[List[Optional[torch.Tensor]]] = [[None] * self.group for _ in range(self.depth)]
        self.pending: List[tuple] = []           # (indices, completion event, inputs kept alive until then)
        self.last_done: List[Optional[torch.cuda.Event]] = [None] * self.depth    # completion of a slot's latest launch
        self.retired: List[tuple] = []           # (outgrown scratch buffer, the event it may still be in use until)
        self.releases: List[tuple] = []          # (indices, callback) waiting for their decode's launch (when_done)
        self.waiting: List[tuple] = []          # batches collected for the next group
        self.turn = 0

    def _scratch(self, slot, k, nbytes):
        """Scratch buffer k of `slot`, at least `nbytes`: allocated on (and only ever used on) the slot's side stream, so
        the caching allocator's stream bookkeeping is right; a buffer that is outgrown stays alive until the slot's
        latest launch -- which may still be reading its history -- has completed."""
        buf = self.scratch[slot][k]
        if buf is None or buf.numel() < nbytes:
            if buf is not None and self.last_done[slot] is not None:
                self.retired.append((buf, self.last_done[slot]))
            self.retired = [entry for entry in self.retired if not entry[1].query()]
            with torch.cuda.stream(self.streams[slot]):
                buf = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
            self.scratch[slot][k] = buf
        return buf

    def reserve(self, batch: int, frames: int, states: int) -> None:
        """Allocate every slot's scratch for `group` batches of this shape now, so that no decode pays for a
        device allocation later (a serving loop calls this once; bench.py does before it starts the clock)."""
        need = viterbi.workspace_bytes(batch, frames, states)
        for slot in range(self.depth):
            for k in range(self.group):
                self._scratch(slot, k, need)

    def when_done(self, indices: torch.Tensor, release: Callable) -> None:
        """Call `release(event)` once the decode that produces `indices` has been LAUNCHED, with its completion event
        (the many-file driver returns a batch's staging buffer to its pool this way: torbi_amd/slabs.py)."""
        for entry in self.pending:
            if entry[0] is indices:           # launched already (group == 1, or a group that has just been flushed)
                release(entry[1])
                return
        self.releases.append((indices, release))

    def decode(self, observation, batch_frames, transition, initial, after: Optional[Callable] = None) -> torch.Tensor:
        """Enqueue one decode; arguments as `torbi_amd.decode` (tensors on `self.device`).

        `after(indices)` (optional) runs on the same side stream right after the decode, e.g. the
        RCCL gather of a sharded batch or an asynchronous copy to pinned host memory; its return value
        replaces the indices handed back here only for `group` == 1.
        """
        # the inputs are ready where the caller's current stream is NOW (a launch group is launched later, possibly
        # from another stream context: it waits for this event, not for whatever is current then)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        if self.group == 1:
            return self._launch([(observation, batch_frames, transition, initial, after, None, ready)])[0]
        B, T, _ = observation.shape
        indices = torch.empty((B, T), dtype=torch.int32, device=self.device)
        # a group shares one transition / initial: a different model flushes what has been collected
        if self.waiting and (self.waiting[0][2] is not transition or self.waiting[0][3] is not initial
                             or self.waiting[0][0].shape[-1] != observation.shape[-1]):
            self.flush()
        self.waiting.append((observation, batch_frames, transition, initial, after, indices, ready))
        if len(self.waiting) >= self.group:
            self.flush()
        return indices

    def _ready_prefix(self) -> int:
        """How many of the collected batches, from the oldest on, have their inputs ready on the device."""
        count = 0
        for batch in self.waiting:
            if not batch[6].query():
                break
            count += 1
        return count

    def _launch_first(self, count: int) -> None:
        if count > 0:
            batches, self.waiting = self.waiting[:count], self.waiting[count:]
            self._launch(batches)

    def flush_if_idle(self) -> bool:
        """Launch the collected batches whose inputs are ready if the device has nothing else to do (every earlier
        launch has completed).  A job whose batches arrive more slowly than they are decoded (files crossing the host
        link: 26-46 ms per batch against 5-10 ms of decode) then decodes every batch as it arrives instead of waiting
        for a full group -- nothing is left to decode behind the last copy but the last batch -- while a job with
        everything resident still fills its groups."""
        if not self.waiting:
            return False
        self._prune()
        if self.pending:
            return False
        count = self._ready_prefix()
        self._launch_first(count)
        return count > 0

    def flush(self) -> None:
        """Launch the batches collected so far (a partial group)."""
        if self.waiting:
            batches, self.waiting = self.waiting, []
            self._launch(batches)

    def _launch(self, batches):
        slot = self.turn % self.depth
        self.turn += 1
        stream = self.streams[slot]
        transition, initial = batches[0][2], batches[0][3]
        # the inputs may have been produced on other streams: wait for each batch's readiness event
        for batch in batches:
            stream.wait_event(batch[6])
        results = []
        with torch.cuda.stream(stream):
            if len(batches) == 1 and batches[0][5] is None:
                observation, batch_frames, _, _, after, _, _ = batches[0]
                B, T, S = observation.shape
                scratch = self._scratch(slot, 0, viterbi.workspace_bytes(B, T, S))
                indices = viterbi.decode(observation, batch_frames, transition, initial, workspace=scratch,
                                         reuse_preparation=self.reuse_preparation, path=self.path)
                if after is not None:
                    indices = after(indices)
                results.append(indices)
            else:
                spaces = [self._scratch(slot, k, viterbi.workspace_bytes(*b[0].shape)) for k, b in enumerate(batches)]
                # consecutive launch groups alternate between shortest-tiles-first and longest-tiles-first: the
                # long workgroups of one group then start on the CUs the short ones of the other have just left
                # (4 ragged groups over two streams: 37.1 ms against 39.5 ms all longest-first)
                decoded = viterbi.decode_batches([b[0] for b in batches], [b[1] for b in batches], transition,
                                                 initial, workspaces=spaces,
                                                 reuse_preparation=self.reuse_preparation, path=self.path,
                                                 out=[b[5] for b in batches],
                                                 shortest_first=self.depth > 1 and self.turn % 2 == 1)
                for (_, _, _, _, after, _, _), indices in zip(batches, decoded):
                    if after is not None:
                        after(indices)
                    results.append(indices)
            done = torch.cuda.Event()
            done.record(stream)
            self.last_done[slot] = done
        # Inputs and indices were allocated on the caller's stream and are used on this one: `pending` holds them until
        # `done` has completed, so the allocator cannot hand their memory out early.  (record_stream() would do the
        # same by deferring every free behind an event; on a job that frees a batch per decode that cost the
        # many-file benchmark a third of its rate.)
        self._prune()
        keep = [(b[0], b[1]) for b in batches] + [(transition, initial)]
        for indices in results:
            self.pending.append((indices, done, keep))
        if self.releases:
            launched = {id(indices) for indices in results}
            for indices, release in [entry for entry in self.releases if id(entry[0]) in launched]:
                release(done)
            self.releases = [entry for entry in self.releases if id(entry[0]) not in launched]
        return results

    def _prune(self) -> None:
        """Forget the decodes that have completed (and let go of their inputs)."""
        self.pending = [entry for entry in self.pending if not entry[1].query()]

    def wait(self, indices: torch.Tensor) -> torch.Tensor:
        """Block the host until the decode that produced `indices` has finished."""
        for k, entry in enumerate(self.waiting):
            if entry[5] is indices:
                # still being collected: launch it now, together with everything before it and whatever else is ready
                # (not the batches whose inputs are still on their way: they would hold this one's launch back)
                self._launch_first(max(k + 1, self._ready_prefix()))
                break
        for entry in self.pending:
            if entry[0] is indices:
                entry[1].synchronize()
                break
        self._prune()
        return indices

    def synchronize(self) -> None:
        """Block the host until every enqueued decode has finished."""
        self.flush()
        for entry in self.pending:
            entry[1].synchronize()
        self.pending = []
        for stream in self.streams:
            stream.synchronize()
