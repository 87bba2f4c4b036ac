"""Chunked (approximate) decoding of long sequences: the host-side mirror of reference torbi/chunk.py:12-85.

A long sequence is cut where two ADJACENT frames are both near-certain (normalised entropy below a threshold):
there the most likely path is pinned by the observations, so decoding the pieces separately and joining them
changes little.  The pieces become extra batch rows (`data.collate`) and are joined again after decoding
(`data.separate`), which turns one latency-bound long decode into batch parallelism -- the regime the MI355X
kernels want (a (1, 8000, S) item occupies one 16-item tile; forty 200-frame pieces fill three).

Off by default like upstream (`core.MIN_CHUNK_SIZE = None`, reference torbi/config/defaults.py:41); it is an
approximation: results can differ from the unchunked decode, and parity is "same cut points and same joined
indices as the reference's Python" (tests/golden/golden_api.npz), not oracle equality.

The entropy is evaluated with the reference's own expression on the host (torch CPU ops, same operand layout), so
frames that sit near the threshold fall on the same side as upstream.
"""
from typing import List, Optional

import torch


def entropy(observation: torch.Tensor) -> torch.Tensor:
    """Entropy of every frame of a (frames, states) tensor of natural-log probabilities, divided by log(states)
    (reference chunk.py:81-85; evaluated state-major like upstream so the summation order is the same)."""
    by_state = observation.T
    weighted = torch.exp(by_state) * by_state
    return -(weighted.sum(dim=0) / torch.log(torch.tensor(by_state.shape[0])))


def split(observation: torch.Tensor, min_chunk_size: int, entropy_threshold: float) -> List[int]:
    """Frames at which `observation` (frames, states) is cut (reference chunk.py:57-78): scanning from
    `min_chunk_size`, the first frame that is low-entropy together with its predecessor ends a piece, and the
    scan resumes `min_chunk_size` frames later.  A NaN entropy (a -inf log-probability) never qualifies."""
    if min_chunk_size is None or int(min_chunk_size) < 1:
        raise ValueError('min_chunk_size must be a positive integer')
    step = int(min_chunk_size)
    low = entropy(observation) < entropy_threshold
    # frames i >= 1 with low[i] and low[i-1], ascending
    eligible = torch.nonzero(low[1:] & low[:-1]).flatten() + 1
    cuts, at = [], step
    frames = observation.shape[0]
    while at < frames:
        where = int(torch.searchsorted(eligible, at))
        if where >= eligible.numel():
            break
        cut = int(eligible[where])
        cuts.append(cut)
        at = cut + step
    return cuts


def chunk(observation: torch.Tensor, min_chunk_size: Optional[int] = None,
          entropy_threshold: Optional[float] = None) -> List[torch.Tensor]:
    """The pieces of a (frames, states) observation between its cut points (reference chunk.py:12-48); their
    concatenation is the input.  Defaults come from `core.MIN_CHUNK_SIZE` / `core.ENTROPY_THRESHOLD` at call
    time."""
    from . import core
    if min_chunk_size is None:
        min_chunk_size = core.MIN_CHUNK_SIZE
    if entropy_threshold is None:
        entropy_threshold = core.ENTROPY_THRESHOLD
    bounds = [0] + split(observation, min_chunk_size, entropy_threshold) + [observation.shape[0]]
    return [observation[lo:hi] for lo, hi in zip(bounds[:-1], bounds[1:])]
