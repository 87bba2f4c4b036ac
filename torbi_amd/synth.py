"""Deterministic, platform-independent synthetic inputs for parity tests and bench.py.

A counter-based integer hash (the splitmix64 finaliser) turns (stream, flat index) into a
24-bit integer u; the value is 0 - u * 2**-20, an fp32 that is exact by construction
(24-bit integer times a power of two), i.e. an unnormalised log-score in (-16, 0].  No
log/exp/softmax is involved, so the bits are identical on every machine and on the GPU
(torbi_hip_fill_synthetic in csrc/ computes the same function on device).

Inputs produced here are fed at the operator boundary (`decode`), bypassing the epsilon
round trip of `from_probabilities` (reference torbi/core.py:193-197), because CPU and GPU
exp/log differ in the last ulp.

Streams: observation = 1, transition = 2, initial = 3.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
C1 = np.uint64(0xBF58476D1CE4E5B9)
C2 = np.uint64(0x94D049BB133111EB)

STREAM_OBSERVATION = 1
STREAM_TRANSITION = 2
STREAM_INITIAL = 3


def hash_u24(stream, start, count, seed=0):
    """24-bit hash values for flat indices start .. start+count-1 of a stream."""
    with np.errstate(over='ignore'):
        idx = np.arange(start, start + count, dtype=np.uint64)
        z = idx + (np.uint64(stream) + np.uint64(seed) * np.uint64(1000003)) * GOLDEN
        z = (z ^ (z >> np.uint64(30))) * C1
        z = (z ^ (z >> np.uint64(27))) * C2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.uint32)


def scores(stream, shape, seed=0, start=0, chunk=1 << 24):
    """fp32 array of `shape` holding -(u24 * 2**-20) for consecutive flat indices."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        u = hash_u24(stream, start + lo, hi - lo, seed)
        out[lo:hi] = np.float32(0) - u.astype(np.float32) * np.float32(2.0 ** -20)
    return out.reshape(shape)


def problem(B, T, S, seed=0):
    """(observation, transition, initial) numpy arrays for a dense synthetic problem."""
    return (scores(STREAM_OBSERVATION, (B, T, S), seed),
            scores(STREAM_TRANSITION, (S, S), seed),
            scores(STREAM_INITIAL, (S,), seed))


def lengths(count, lo, hi, seed=0, stream=7):
    """Deterministic integer lengths in [lo, hi] (many-file workload)."""
    u = hash_u24(stream, 0, count, seed).astype(np.int64)
    return (lo + (u % (hi - lo + 1))).astype(np.int32)


def banded_transition(S, half_width, dtype=np.float32, tiny=False):
    """Log of the triangular banded transition the reference's evaluation builds
    (torbi/evaluate/core.py:24-33): clip(w - |x - y|, 0) row-normalised; log(0) = -inf
    outside the band.  `half_width` plays the role of max_bins_per_frame.
    tiny: log(p + tiny) in float32 instead -- what the operator really sees when the evaluation calls
    from_files_to_files(transition_file=..., log_probs=True) (torbi/core.py:341-347): log(tiny) = -87.34 outside the band."""
    x = np.arange(S)
    tri = np.clip(half_width - np.abs(x[:, None] - x[None, :]), 0, None).astype(np.float64)
    tri = tri / tri.sum(axis=1, keepdims=True)
    if tiny:
        p = tri.astype(np.float32)
        return np.log(p + np.finfo(np.float32).tiny).astype(dtype)
    with np.errstate(divide='ignore'):
        return np.log(tri).astype(dtype)
