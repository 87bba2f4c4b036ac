"""Operator wrapper: the host-side mirror of reference torbi/viterbi.py:5-53.

`decode` keeps the reference's name, argument order and meaning, and calls the MI355X HIP
implementation through the C ABI (include/torbi_hip.h) instead of
`torch.ops.torbi.viterbi_decode` (reference torbi/csrc/ops.cpp:17).
"""
import threading
import ctypes
import os
from typing import Optional

import torch

from . import _lib, state
from .state import _version_of


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            'torbi_amd.decode needs a HIP device (PyTorch-ROCm reports none). This package has '
            'no CPU fallback (decode_cpu is the CPU operator); the reference CPU implementation is only restated under '
            'oracle/ as a test oracle.')


def _check_inputs(observation, batch_frames, transition, initial):
    if observation.dim() != 3:
        raise RuntimeError(
            f'observation must have shape (batch, frames, states); got {tuple(observation.shape)}')
    B, T, S = observation.shape
    if T < 1 or S < 1:
        raise RuntimeError('observation needs at least one frame and one state')
    # dtype errors mirror what ATen's data_ptr<T>() raises in the reference operator
    # (torbi/csrc/cuda/viterbi.cu:203-215): fp32 scores, int32 lengths
    for name, tensor, dtype in (('observation', observation, torch.float32),
                                ('transition', transition, torch.float32),
                                ('initial', initial, torch.float32),
                                ('batch_frames', batch_frames, torch.int32)):
        if tensor.dtype != dtype:
            raise RuntimeError(
                f'expected scalar type {dtype} for {name} but found {tensor.dtype}')
    if tuple(batch_frames.shape) != (B,):
        raise RuntimeError(f'batch_frames must have shape ({B},); got {tuple(batch_frames.shape)}')
    if tuple(transition.shape) != (S, S):
        raise RuntimeError(f'transition must have shape ({S}, {S}); got {tuple(transition.shape)}')
    if tuple(initial.shape) != (S,):
        raise RuntimeError(f'initial must have shape ({S},); got {tuple(initial.shape)}')
    return B, T, S


def workspace_bytes(batch: int, frames: int, states: int) -> int:
    """Scratch bytes one decode of this shape needs (trellis + posterior rows)."""
    return int(_lib.load().torbi_hip_workspace_bytes(batch, frames, states))


def _path_flag(path: str) -> int:
    """TORBI_HIP_PATH_FLAG(path): the forward path travels with the call, nothing process-wide is touched."""
    return (FORWARD_PATHS[path] + 1) << 4


def _resolve_path(trans, transition, B, S, device, path, tiles, count=1, items=None):
    """The path name this call passes to the library."""
    forced = _forced_path if path is None else path
    if forced not in FORWARD_PATHS:
        raise ValueError(f'forward path must be one of {sorted(FORWARD_PATHS)}; got {forced!r}')
    if forced not in ('auto', 'band'):
        return forced
    index = torch.device(device).index or 0
    total = B if items is None else items
    small = 2 <= S <= SMALL_STATES and forward_path(total, S, 'auto', index) == 'small'
    if not (small and forced == 'auto'):
        # a banded matrix (the reference's pitch model, torbi/evaluate/core.py:24-33) whose band the band kernel covers
        # (csrc/band_forward.hpp): every finite cell and no other, the time loop inside one launch.  AUTO leaves the
        # handful of sequences the held-matrix kernel decodes to it.
        if _band_of(trans, transition, S, total, index) is not None \
                and (forced == 'band' or not (count == 1 and forward_path(B, S, 'auto', index) == 'held')):
            return 'band'
    if small or forced == 'band':
        return 'auto'               # one wavefront / workgroup per sequence, whatever the matrix looks like (csrc/small_states.hpp)
    chosen = _choose_path(trans, transition, B, S)
    banded = chosen == 'dense'
    cus = compute_units(device)
    group_like = 2 * tiles > cus or (count > 1 and B > 16)
    single = count == 1 and B > 16 and not banded
    if 64 <= S <= 4096 and tiles <= 16384 and (group_like or single):
        # The time-resident kernel (csrc/resident_forward.hpp), whatever the per-timestep choice would be:
        #  * enough items to give more than half the compute units a workgroup of 16 each: whole tiles per workgroup --
        #    also for a narrow band (its lists end at the band edge, so the scan is short whatever the posteriors look
        #    like: 58 M against 32 M timesteps/s for eight batches of peaked rows with the pitch transition);
        #  * a launch group that fills less (the tail of a job, a small job): tiles split over clusters of workgroups
        #    (2 x 512 items: 25.4 us per timestep against 62.7 with whole tiles and 2 x 20.4 one batch after the other);
        #  * ONE batch of more than 16 items: clusters -- 12.8 against 14.5 us per timestep at 17 items, 15.6 against 20.2
        #    at 512, 20.4 against 34.9 at 768 (profiles/r03_cluster_sweep_one_poller.txt), 128 x 2000 x 4096 in 54.2
        #    against 55.9 ms on the per-timestep kernel (profiles/r03_bench.json) -- except for a narrow band: the dense
        #    kernel's -inf skipping costs by the item (80 pieces of a chunked sequence: 1.9 against 3.0 ms; 512 peaked
        #    rows 8.0 against 8.3 ms), a cluster timestep has a floor of ~12 us however few items it carries;
        # unless the scan statistics of an earlier time-resident launch with this matrix say that hardly anything is
        # pruned (see _watch_resident).  On peaked rows with a dense matrix the time-resident kernel still runs at the
        # dense kernel's rate or better (35-38 against 38-40 us per timestep for one batch, twice its rate in launch groups).
        # 'cluster' lets the library pick the form (whole tiles once 2 * tiles > compute units).
        if banded or chosen in ('pruned', 'dense'):
            losing = not banded and _resident_is_losing(transition, S, single=not group_like)
            chosen = 'dense' if losing else 'cluster'
    return chosen


# fraction of a row's S/16 list blocks per scan above which the dense kernel wins (tools/peaked_group_probe.py at 1440
# states: 11 blocks -> 53 M timesteps/s, 19 -> 36 M, 31 -> 23 M, 44 -> 17 M; the dense kernel 12.7 M whatever the data)
RESIDENT_GATE = 0.65


# one batch in clusters against the dense kernel (tools/depth_probe.py, 512 x 500 x 1440: 18.7 us per timestep at 0.12 of
# a row's blocks, 42.3 at 0.46; the dense kernel 40.4 whatever the data): the crossing is near 0.43
SINGLE_BATCH_GATE = 0.43
# a launch that kept ONE seed per item walks further than one with three before the bound bites (0.60 against 0.46 of a
# row on peaked rows, 0.130 against 0.119 on the benchmark's): depths measured that way are compared with gates this
# much higher
ONE_SEED_DEPTH = 1.25


# AUTO's data-dependent gates follow the DATA, not the first call: every time-resident launch AUTO chose leaves a sample of
# its scan depth (copied to pinned memory behind the decode, folded in by a later call without waiting), the depth a gate
# sees is an exponentially weighted mean that leans on the newest sample, and while the gates keep a matrix on the dense
# kernel -- which leaves no statistics -- every DENSE_PROBE_EVERY-th call takes the time-resident kernel anyway to look again.
# Peaked batches followed by flat ones (or the reverse) with ONE matrix change route within three calls.
DEPTH_NEWEST_WEIGHT = 0.75
DENSE_PROBE_EVERY = 3           # ... the first look again; the interval doubles while the looks keep saying "losing"
DENSE_PROBE_AT_MOST = 48        # (flat data for good: one call in 48 pays for the look, not one in three)
_depth_lock = threading.Lock()  # host threads that share one transition tensor share its record


def _capturing() -> bool:
    """The current stream is being captured into a HIP graph (torch.cuda.graph): nothing that waits for or asks about
    the device may run."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _depth_record(transition: torch.Tensor, states: int):
    """[scan depth of time-resident launches with this matrix in list blocks per wave pass, on the scale of a three-seed
    launch (None until known), pending sample (pinned stats, event, seeds per item of that launch) or None, calls between
    two looks while the gates keep the matrix on the dense kernel, the pinned buffer kept for the samples, consecutive calls
    the gates sent to the dense kernel], kept with the tensor's notes (torbi_amd/state.py); read and written under
    `_depth_lock`."""
    kept = state.notes(transition)
    if kept is None:
        return None
    return kept.setdefault(('depth', states), [None, None, DENSE_PROBE_EVERY, None, 0])


def _known_depth(transition: torch.Tensor, states: int):
    """Blocks per scan on the scale of a three-seed launch, or None while no statistics have come back (never waits)."""
    known = _depth_record(transition, states)
    if known is None:
        return None
    with _depth_lock:
        pending = known[1]
        # (no event query while the stream is being captured into a HIP graph: it would invalidate the capture; the
        # graph then replays the route that what is known so far chooses)
        if pending is not None and not _capturing() and pending[1].query():
            stats, _, seeds = pending
            if int(stats[64:120].sum()) > 0:
                sample = critical_blocks(stats) / (ONE_SEED_DEPTH if seeds == 1 else 1.0)
                known[0] = sample if known[0] is None else DEPTH_NEWEST_WEIGHT * sample + (1.0 - DEPTH_NEWEST_WEIGHT) * known[0]
            known[1] = None             # (only now may another thread start a copy into the pinned buffer)
        return known[0]


def _resident_is_losing(transition: torch.Tensor, states: int, single: bool = False) -> bool:
    """Time-resident launches with this matrix walk so many list blocks per scan that the dense kernel is faster
    (flat or nearly flat rows: nothing to prune).  Read without blocking from the statistics the kernel leaves.
    `single`: the question is asked for ONE batch below half the chip (clusters against one dense batch).  Every
    DENSE_PROBE_EVERY-th consecutive "yes" is answered "no": the dense kernel leaves no statistics, and data that have become
    prunable again would otherwise never be seen."""
    depth = _known_depth(transition, states)
    losing = depth is not None and depth > (SINGLE_BATCH_GATE if single else RESIDENT_GATE) * states / 16.0
    known = _depth_record(transition, states)
    if known is not None:
        with _depth_lock:
            if not losing:
                known[4], known[2] = 0, DENSE_PROBE_EVERY
            else:
                known[4] += 1
                if known[4] >= known[2]:            # look again; the next look twice as far away while this keeps happening
                    known[4] = 0
                    known[2] = min(2 * known[2], DENSE_PROBE_AT_MOST)
                    return False
    return losing


# scans this shallow (fraction of a row's S/16 list blocks per wave pass; 0.12 on the benchmark, 0.28-0.47 on peaked rows
# with a dense matrix) gain nothing from three seeds per item: one is as fast and reads a third of the transposed matrix
FEW_SEEDS_GATE = 0.17


def _seed_flag(transition: torch.Tensor, states: int) -> int:
    """TORBI_HIP_FEW_SEEDS (512) / TORBI_HIP_MANY_SEEDS (1024) for a time-resident launch with this matrix, or 0 (the
    library's default: three seeds per item with whole tiles, one in the cluster form) while nothing is known: once an
    earlier launch's scan statistics are in (read without blocking) shallow scans select one seed, deep ones three."""
    depth = _known_depth(transition, states)
    if depth is None:
        return 0
    return 512 if depth <= FEW_SEEDS_GATE * states / 16.0 else 1024


def _seeds_kept(flags: int, chosen: str, tiles: int, device) -> int:
    """Seeds per item of a time-resident launch with these flags (csrc/torbi_hip.hip few_seeds())."""
    if flags & 512:
        return 1
    if flags & 1024:
        return 3
    return 1 if chosen != 'resident' and 2 * tiles <= compute_units(device) else 3


def _few_seeds(transition: torch.Tensor, states: int) -> bool:
    return _seed_flag(transition, states) == 512


def _watch_resident(transition, workspace, batch, frames, states, seeds=3) -> None:
    """After a time-resident launch chosen by AUTO: copy the scan statistics it leaves in its first workspace to pinned
    host memory (asynchronously; folded in by a later call, never waited for) -- unless the previous sample is still on its
    way.  `seeds`: what the launch kept per item."""
    known = _depth_record(transition, states)
    if known is None or _capturing():       # (a sample copied by a graph's replays would never be looked at: its event is a node)
        return
    with _depth_lock:
        if known[1] is not None:
            return
        if known[3] is None:
            known[3] = torch.empty((128,), dtype=torch.int32, pin_memory=True)
        known[3].copy_(scan_stats(workspace, batch, frames, states, path='resident'), non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(workspace.device))
        known[1] = (known[3], done, seeds)


def band_reach(trans: torch.Tensor, original: torch.Tensor, states: int):
    """(reach_left, reach_right) of a transition matrix the band kernel could take -- transition[j][i] is -inf unless
    j - reach_left <= i <= j + reach_right (include/torbi_hip.h, torbi_hip_band_reach) -- else None.  One small kernel and
    a host sync the first time a tensor version is seen; kept with the tensor's notes (torbi_amd/state.py)."""
    if states % 4 or not 64 <= states <= BAND_MAX_STATES or not trans.is_cuda or trans.data_ptr() % 16:
        return None
    found = _device_reach(trans, original, states)
    found = (max(found[0], 0), max(found[1], 0))              # (-1, -1: no finite entry at all -- a band of the diagonal)
    return found if found[0] + found[1] + 4 <= BAND_MAX_WINDOW else None


def band_over(trans: torch.Tensor, original: torch.Tensor, states: int):
    """(reach_left, reach_right, background) of a transition matrix that holds ONE value outside a band -- the reference's
    evaluation decodes with log(p + tiny), log(tiny) = -87.34 outside the pitch band, not -inf (torbi/evaluate/core.py:97-103 ->
    torbi/core.py:341-347) --: `background` = the corner entry transition[0][S - 1], the reaches over every other value
    (include/torbi_hip.h, torbi_hip_band_reach_over); None where the band kernels could not take it.  A matrix that is -inf
    outside its band answers (reach_left, reach_right, -inf).  One small kernel and a host sync per tensor version."""
    if states % 4 or not 64 <= states <= BAND_MAX_STATES or not trans.is_cuda or trans.data_ptr() % 16:
        return None
    found = _device_look(trans, original, states)
    left, right, background = max(found[0], 0), max(found[1], 0), found[2]
    if left + right + 4 > BAND_MAX_WINDOW or background != background or background == float('inf'):
        return None
    return left, right, background


def _band_of(trans, transition, S, items, index):
    """What the band route would be told about this matrix for `items` sequences: (reach_left, reach_right, background), or
    None when the band kernels do not cover it.  -inf outside the band first (both forms of the kernel), then ONE finite
    constant outside it (whole tiles only)."""
    lib = _lib.load()
    reach = band_reach(trans, transition, S)
    if reach is not None and lib.torbi_hip_band_members(int(items), S, reach[0], reach[1], index) > 0:
        return reach[0], reach[1], float('-inf')
    over = band_over(trans, transition, S)
    if over is not None and over[2] != float('-inf') \
            and lib.torbi_hip_band_members_over(int(items), S, over[0], over[1], ctypes.c_float(over[2]), index) > 0:
        return over
    return None


def _device_look(trans: torch.Tensor, original: torch.Tensor, states: int):
    """(reach_left, reach_right, background) of a device matrix as torbi_hip_band_reach_over answers -- `background` = the
    corner entry transition[0][S - 1], the reaches over every entry that differs from it bit for bit --, looked at ONCE per
    tensor version (one small kernel + a host sync) for both questions: -inf outside a band, or one constant."""
    kept = state.notes(original)
    found = kept.get(('band_over', states)) if kept is not None else None
    if found is None and torch.cuda.is_current_stream_capturing():
        # (the look synchronises the stream: not while a graph is being captured -- round-5 advisor.  Unknown = not banded:
        # the routes that do not need a band take the call; look once before capturing to have the band kernels in the graph)
        return states - 1, states - 1, 0.0
    if found is None:
        left, right, background = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_float(0.0)
        _lib.check(_lib.load().torbi_hip_band_reach_over(trans.data_ptr(), states, trans.device.index or 0,
                                                         ctypes.c_void_p(torch.cuda.current_stream(trans.device).cuda_stream),
                                                         ctypes.byref(left), ctypes.byref(right), ctypes.byref(background)),
                   'torbi_hip_band_reach_over')
        found = (left.value, right.value, background.value)
        if kept is not None:
            kept[('band_over', states)] = found
    return found


def _device_reach(trans: torch.Tensor, original: torch.Tensor, states: int):
    """(reach_left, reach_right) of a device matrix with -inf as "outside" (torbi_hip_band_reach's answer): a matrix whose
    corner is not -inf has an entry that is not -inf at distance S - 1."""
    left, right, background = _device_look(trans, original, states)
    return (left, right) if background == float('-inf') else (states - 1, states - 1)


def tiles_of(batch: int, states: int) -> int:
    """Tiles of a batch in the time-resident kernel: 16 items each up to 2048 states, 8 above (the posterior tile has to
    fit the 160 KB LDS; csrc/resident_forward.hpp)."""
    items = 16 if states <= 2048 else 8
    return (batch + items - 1) // items


def compute_units(device) -> int:
    """Compute units of a HIP device as the library's tiling plans see them."""
    index = torch.device(device).index
    index = torch.cuda.current_device() if index is None else index
    known = _compute_units.get(index)
    if known is None:
        known = int(_lib.load().torbi_hip_compute_units(index))
        if known <= 0:
            raise RuntimeError(f'torbi_hip_compute_units({index}) failed with code {known}')
        _compute_units[index] = known
    return known


_compute_units = {}


def decode(
    observation: torch.Tensor,
    batch_frames: torch.Tensor,
    transition: torch.Tensor,
    initial: torch.Tensor,
    num_threads: Optional[int] = 0,
    workspace: Optional[torch.Tensor] = None,
    reuse_preparation: bool = False,
    path: Optional[str] = None,
    _profile: Optional[list] = None,
) -> torch.Tensor:
    """Decode a time-varying categorical distribution (log space) on an MI355X

    Args:
        observation: :math:`(N, T, S)` float32 log-probabilities
        batch_frames: :math:`(N)` int32 sequence length of each batch item
        transition: :math:`(S, S)` float32 log transition matrix, indexed [next, prev]
            (reference torbi/csrc/viterbi.cpp:81-86)
        initial: :math:`(S)` float32 log initial distribution
        num_threads: accepted for signature compatibility (reference torbi/viterbi.py:51-52
            sets the CPU thread count); ignored on the GPU (`decode_cpu` takes it)
        reuse_preparation: with `workspace`, a promise that nothing else has written to the
            workspace since the previous decode that used it; when that decode had the same shape,
            forward path and transition tensor (same storage, same version, same stream) the
            per-transition preparation (sorted rows / packed panels, ~0.2 ms at 1440 states) is
            taken from the workspace instead of being rebuilt (include/torbi_hip.h,
            TORBI_HIP_REUSE_TRANSITION)
        workspace: optional uint8 scratch tensor on the compute device with at least
            `workspace_bytes(N, T, S)` bytes; allocated from torch's caching allocator if None
        path: forward recurrence for THIS call ('auto', 'dense', 'pruned', 'resident', 'cluster', 'held', 'band'; None =
            the process default of `set_forward_path`).  Every path returns the same indices.

    Return:
        indices: :math:`(N, T)` int32 decoded bin indices, on the device of `observation`

    Tensors on the CPU are moved to the current HIP device, decoded there, and the indices
    are returned on the CPU (the reference returns on the input device as well).  Calls from
    different host threads (one per stream or device) are independent.
    """
    B, T, S = _check_inputs(observation, batch_frames, transition, initial)
    _require_gpu()
    lib = _lib.load()

    home = observation.device
    device = home if home.type == 'cuda' else torch.device('cuda', torch.cuda.current_device())
    obs = observation.to(device).contiguous()
    frames = batch_frames.to(device).contiguous()
    trans = transition.to(device).contiguous()
    init = initial.to(device).contiguous()

    indices = torch.empty((B, T), dtype=torch.int32, device=device)
    if B == 0:
        return indices.to(home)
    need = lib.torbi_hip_workspace_bytes(B, T, S)
    own_scratch = workspace is None
    if own_scratch:
        workspace = torch.empty((need,), dtype=torch.uint8, device=device)
    elif (workspace.device != device or workspace.dtype != torch.uint8
          or workspace.numel() < need or not workspace.is_contiguous()):
        raise RuntimeError(f'workspace must be a contiguous uint8 tensor of >= {need} bytes on {device}')

    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(device).cuda_stream
    chosen = _resolve_path(trans, transition, B, S, device, path, tiles_of(B, S))
    args = (obs.data_ptr(), frames.data_ptr(), trans.data_ptr(), init.data_ptr(),
            indices.data_ptr(), workspace.data_ptr(), workspace.numel(), B, T, S, index,
            ctypes.c_void_p(stream))
    flags = _path_flag(chosen)
    if _reusable(workspace, transition, (B, T, S, chosen, stream), reuse_preparation):
        flags |= 1                                  # TORBI_HIP_REUSE_TRANSITION
    if chosen in TIME_RESIDENT:
        flags |= _seed_flag(transition, S)          # TORBI_HIP_FEW_SEEDS / _MANY_SEEDS once the scan depth is known
    kept = _kept_preparation(transition, B, S, chosen, device, index) if own_scratch and chosen != 'band' else None
    if chosen == 'band':
        one = (_lib.Batch * 1)(_lib.Batch(obs.data_ptr(), frames.data_ptr(), indices.data_ptr(), workspace.data_ptr(),
                                          workspace.numel(), B, T))
        left, right, background = _band_of(trans, transition, S, B, index)
        phases = (ctypes.c_float * 6)() if _profile is not None else None
        _lib.check(lib.torbi_hip_viterbi_decode_banded_over(one, 1, trans.data_ptr(), init.data_ptr(), S, left, right,
                                                            ctypes.c_float(background), index, ctypes.c_void_p(stream), flags, phases),
                   'torbi_hip_viterbi_decode_banded_over')
        if _profile is not None:
            _profile[:] = list(phases)
    elif kept is not None:       # the per-call scratch is new every time; the preparation stays with the matrix
        one = (_lib.Batch * 1)(_lib.Batch(obs.data_ptr(), frames.data_ptr(), indices.data_ptr(), workspace.data_ptr(),
                                          workspace.numel(), B, T))
        phases = (ctypes.c_float * 6)() if _profile is not None else None
        _lib.check(kept.call(device, lambda pointer, size, reuse, filled: lib.torbi_hip_viterbi_decode_batches_prepared(
            one, 1, trans.data_ptr(), init.data_ptr(), S, index, ctypes.c_void_p(stream), flags | reuse, phases,
            pointer, size, filled)), 'torbi_hip_viterbi_decode_batches_prepared')
        if _profile is not None:
            _profile[:] = list(phases)
    elif _profile is None:
        _lib.check(lib.torbi_hip_viterbi_decode_ex(*args, flags), 'torbi_hip_viterbi_decode_ex')
    else:
        phases = (ctypes.c_float * 6)()
        _lib.check(lib.torbi_hip_viterbi_decode_profiled(*args, flags, phases),
                   'torbi_hip_viterbi_decode_profiled')
        _profile[:] = list(phases)
    if chosen in TIME_RESIDENT and (_forced_path if path is None else path) == 'auto':
        _watch_resident(transition, workspace, B, T, S, _seeds_kept(flags, chosen, tiles_of(B, S), device))
    return indices if home == device else indices.to(home)


def decode_cpu(
    observation: torch.Tensor,
    batch_frames: torch.Tensor,
    transition: torch.Tensor,
    initial: torch.Tensor,
    num_threads: Optional[int] = None,
) -> torch.Tensor:
    """The same operator on the host: the twin of the reference's CPU registration (torbi/csrc/viterbi.cpp:182-234,
    237-239) behind include/torbi_cpu.h, for callers that ask for the CPU (`from_probabilities(gpu=None)`).

    Arguments as `decode`, CPU tensors; `num_threads` OpenMP threads (None / 0 = the runtime's default; the reference
    sets torch's global thread count instead, torbi/viterbi.py:51-52).  Decoded indices are bit-identical to the
    reference CPU operator and to `decode`.  This is an explicit entry point, not a fallback: `decode` never routes here.
    """
    B, T, S = _check_inputs(observation, batch_frames, transition, initial)
    for name, tensor in (('observation', observation), ('batch_frames', batch_frames), ('transition', transition),
                         ('initial', initial)):
        if tensor.device.type != 'cpu':
            raise RuntimeError(f'decode_cpu takes CPU tensors; {name} is on {tensor.device}')
    obs, frames = observation.contiguous(), batch_frames.contiguous()
    trans, init = transition.contiguous(), initial.contiguous()
    indices = torch.empty((B, T), dtype=torch.int32)
    if B == 0:
        return indices
    code = _lib.load_cpu().torbi_cpu_viterbi_decode(obs.data_ptr(), frames.data_ptr(), trans.data_ptr(), init.data_ptr(),
                                                    indices.data_ptr(), B, T, S, int(num_threads or 0))
    if code != 0:
        raise RuntimeError(f'torbi_cpu_viterbi_decode failed with code {code}'
                           + (' (out of memory for the posterior history)' if code == -6 else ''))
    return indices


class _Preparation:
    """The time-resident routes' per-transition preparation kept with the transition tensor (include/torbi_hip.h,
    torbi_hip_viterbi_decode_batches_prepared) for calls that bring no workspace of their own: the reference's calling
    pattern (torbi/core.py:200-206 allocates per call) would otherwise rebuild it every time (0.25 ms at 1440
    states).  Lives in the notes of the tensor at its current version (torbi_amd/state.py), so an in-place change or the
    tensor's death drops it; 25.6 MB at 1440 states."""

    def __init__(self, nbytes, device):
        import threading
        self.buffer = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        self.home = torch.cuda.current_stream(device).cuda_stream      # the stream the caching allocator knows the block by
        self.filled = None            # event recorded behind the call that filled the buffer
        self.stream = None            # the stream that call ran on
        self.lock = threading.Lock()  # held while the filling call is being enqueued

    def call(self, device, run):
        """run(pointer, bytes, reuse_flag, filled) with the buffer ordered behind its filling call on the current stream;
        `filled` is the library's word on whether the buffer holds the preparation afterwards."""
        current = torch.cuda.current_stream(device)
        filled = ctypes.c_int(0)
        if current.cuda_stream != self.home:
            # (also on the filling call: the allocator would otherwise hand the block out again on its home stream while a
            # decode on another stream still reads it, once the tensor's notes -- and with them this object -- are dropped)
            self.buffer.record_stream(current)
        with self.lock:
            if self.filled is None:
                result = run(self.buffer.data_ptr(), self.buffer.numel(), 0, ctypes.byref(filled))
                if result == 0 and filled.value:
                    event = torch.cuda.Event()
                    event.record(current)
                    self.filled, self.stream = event, current.cuda_stream
                return result
        if current.cuda_stream != self.stream:
            current.wait_event(self.filled)
        return run(self.buffer.data_ptr(), self.buffer.numel(), 1, ctypes.byref(filled))


_preparation_lock = __import__('threading').Lock()


def _kept_preparation(transition, B, S, chosen, device, index):
    """The `_Preparation` of `transition` for a call that routes to a time-resident form, else None."""
    if forward_path(B, S, chosen, index) not in TIME_RESIDENT:
        return None
    kept = state.notes(transition)
    if kept is None:
        return None
    key = ('preparation', S, index)
    with _preparation_lock:
        found = kept.get(key)
        if found is None:
            found = kept[key] = _Preparation(int(_lib.load().torbi_hip_preparation_bytes(S)), device)
    return found


def _reusable(workspace, transition, shape_state, wanted) -> bool:
    """Remember what `workspace` will hold after the call being issued (the per-transition preparation for this
    shape, path and stream) and say whether it already does.  The transition is identified by the tensor OBJECT
    and its version (a new tensor can reuse a freed address); a tensor without a version counter (inference
    mode) is never reused.  Kept with the workspace tensor's notes (torbi_amd/state.py)."""
    import weakref
    version = _version_of(transition)
    kept = state.notes(workspace)
    if kept is None:
        return False
    known = kept.get('holds')
    holds = shape_state + (version,)
    hit = (wanted and version is not None and known is not None and known[0] == holds and known[1]() is transition)
    kept['holds'] = (holds, weakref.ref(transition))
    return bool(hit)


def decode_batches(
    observations,
    batch_frames,
    transition: torch.Tensor,
    initial: torch.Tensor,
    workspaces=None,
    reuse_preparation: bool = False,
    path: Optional[str] = None,
    out=None,
    shortest_first: bool = False,
    _profile: Optional[list] = None,
):
    """`decode` for several batches that share `transition` and `initial`, in one call

    The reference decodes a many-file job batch after batch (torbi/core.py:417-457).  Batch items are
    independent, so the batches of such a job can share launches: with enough items (or path='resident')
    the whole group is ONE forward launch -- a workgroup keeps 16 items' posterior rows in its LDS for every
    timestep -- and ONE backtrace launch (include/torbi_hip.h, torbi_hip_viterbi_decode_batches).
    Otherwise the batches are decoded one after the other exactly as `decode` would.

    Args:
        observations: list of (N_k, T_k, S) float32 tensors on one HIP device
        batch_frames: list of (N_k) int32 tensors
        transition, initial: as `decode`
        workspaces: optional list of uint8 scratch tensors, one per batch, each >= workspace_bytes(N_k, T_k, S)
        reuse_preparation: as `decode`, for the first workspace
        path: as `decode`
        out: optional list of preallocated (N_k, T_k) int32 tensors to decode into

    Returns:
        list of (N_k, T_k) int32 index tensors on the device
    """
    count = len(observations)
    if count != len(batch_frames):
        raise RuntimeError('decode_batches needs one batch_frames tensor per observation tensor')
    if count == 0:
        return []
    if count > _lib.MAX_BATCHES:
        raise RuntimeError(f'at most {_lib.MAX_BATCHES} batches per call; got {count}')
    _require_gpu()
    lib = _lib.load()
    device = observations[0].device
    if device.type != 'cuda':
        raise RuntimeError('decode_batches takes tensors that are already on a HIP device')
    S = observations[0].shape[-1]
    shapes = []
    for obs, frames in zip(observations, batch_frames):
        shapes.append(_check_inputs(obs, frames, transition, initial))
        if obs.device != device or frames.device != device or not obs.is_contiguous() or not frames.is_contiguous():
            raise RuntimeError('decode_batches needs contiguous tensors on one device')
    trans = transition.to(device).contiguous()
    init = initial.to(device).contiguous()
    own_scratch = workspaces is None
    if own_scratch:
        workspaces = [torch.empty((lib.torbi_hip_workspace_bytes(B, T, S),), dtype=torch.uint8, device=device)
                      for B, T, _ in shapes]
    if len(workspaces) != count:
        raise RuntimeError('decode_batches needs one workspace per batch')
    if out is None:
        out = [None] * count
    indices = [torch.empty((B, T), dtype=torch.int32, device=device) if given is None else given
               for (B, T, _), given in zip(shapes, out)]
    for (B, T, _), tensor in zip(shapes, indices):
        if (tuple(tensor.shape) != (B, T) or tensor.dtype != torch.int32 or tensor.device != device
                or not tensor.is_contiguous()):
            raise RuntimeError('out tensors must be contiguous int32 (N_k, T_k) tensors on the compute device')
    table = (_lib.Batch * count)()
    for k, (B, T, _) in enumerate(shapes):
        ws = workspaces[k]
        need = lib.torbi_hip_workspace_bytes(B, T, S)
        if ws.device != device or ws.dtype != torch.uint8 or ws.numel() < need or not ws.is_contiguous():
            raise RuntimeError(f'workspace {k} must be a contiguous uint8 tensor of >= {need} bytes on {device}')
        table[k] = _lib.Batch(observations[k].data_ptr(), batch_frames[k].data_ptr(), indices[k].data_ptr(),
                              ws.data_ptr(), ws.numel(), B, T)
    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(device).cuda_stream
    largest = max(B for B, _, _ in shapes)
    tiles = sum(tiles_of(B, S) for B, _, _ in shapes)
    chosen = _resolve_path(trans, transition, largest, S, device, path, tiles, count=count,
                           items=sum(B for B, _, _ in shapes))
    flags = _path_flag(chosen)
    first = next((k for k, (B, _, _) in enumerate(shapes) if B > 0), 0)
    if _reusable(workspaces[first], transition, (tuple(shapes), chosen, stream), reuse_preparation) \
            and (chosen in TIME_RESIDENT or count == 1):
        flags |= 1
    if shortest_first:
        flags |= 256                               # TORBI_HIP_SHORTEST_FIRST
    if chosen in TIME_RESIDENT:
        flags |= _seed_flag(transition, S)         # TORBI_HIP_FEW_SEEDS / _MANY_SEEDS once the scan depth is known
    phases = (ctypes.c_float * 6)() if _profile is not None else None
    kept = _kept_preparation(transition, largest, S, chosen, device, index) if own_scratch and chosen != 'band' else None
    if chosen == 'band':
        left, right, background = _band_of(trans, transition, S, sum(B for B, _, _ in shapes), index)
        _lib.check(lib.torbi_hip_viterbi_decode_banded_over(table, count, trans.data_ptr(), init.data_ptr(), S, left, right,
                                                            ctypes.c_float(background), index, ctypes.c_void_p(stream), flags, phases),
                   'torbi_hip_viterbi_decode_banded_over')
    elif kept is not None:
        _lib.check(kept.call(device, lambda pointer, size, reuse, filled: lib.torbi_hip_viterbi_decode_batches_prepared(
            table, count, trans.data_ptr(), init.data_ptr(), S, index, ctypes.c_void_p(stream), flags | reuse, phases,
            pointer, size, filled)), 'torbi_hip_viterbi_decode_batches_prepared')
    else:
        _lib.check(lib.torbi_hip_viterbi_decode_batches(table, count, trans.data_ptr(), init.data_ptr(), S, index,
                                                        ctypes.c_void_p(stream), flags, phases),
                   'torbi_hip_viterbi_decode_batches')
    if _profile is not None:
        _profile[:] = list(phases)
    if chosen in TIME_RESIDENT and (_forced_path if path is None else path) == 'auto':
        B0, T0, _ = shapes[first]
        _watch_resident(transition, workspaces[first], B0, T0, S, _seeds_kept(flags, chosen, tiles, device))
    return indices


SMALL_STATES = 256                            # small::kBlockMaxS
TIME_RESIDENT = ('resident', 'cluster')       # the two forms of the time-resident kernel (include/torbi_hip.h)
FORWARD_PATHS = {'auto': 0, 'dense': 1, 'pruned': 2, 'resident': 3, 'cluster': 4, 'held': 5, 'band': 6}
BAND_MAX_STATES = 3072                        # 16 members x 192 next-states (csrc/band_forward.hpp)
BAND_MAX_WINDOW = 512                         # band::kMaxWindow: reach_left + reach_right + 4 prev-states per backtrace step


_forced_path = {'d': 'dense', 'p': 'pruned', 'r': 'resident', 'c': 'cluster', 'h': 'held', 'b': 'band'}.get(os.environ.get('TORBI_HIP_FORWARD', 'a')[:1], 'auto')
BANDED_RANGE = 0.25              # rows reaching less than this fraction of the states: dense + -inf skipping


def set_forward_path(path: str = 'auto') -> None:
    """Process-wide DEFAULT of the forward recurrence (include/torbi_hip.h) for calls that do not name one
    (`decode(path=...)`): 'auto' (default), 'dense' (every cell, (max,+) GEMM with -inf block skipping),
    'pruned' (the exact pruned recurrence: the sorted-row scan up to 16 items, the time-resident forms above),
    'resident' / 'cluster' (the pruned recurrence with the time loop inside one launch: whole 16-item tiles per workgroup /
    tiles split over clusters of workgroups) or 'held' (a handful of sequences, the matrix held in registers).  The path itself travels with every call
    (TORBI_HIP_PATH_FLAG), so concurrent decodes from several host threads never see each other's choice.
    Every path returns identical indices; this is a performance knob and a test hook.

    'auto' in this Python layer refines the library's AUTO with one look at the transition matrix, cached per
    tensor version: ONE batch with a matrix whose rows reach only a narrow band of prev-states (mean finite range
    < 25 % of the states, e.g. the reference's pitch transition, torbi/evaluate/core.py:24-33) runs faster on the
    dense kernel's -inf chunk skipping; everything else takes the pruned recurrence.  The look costs one small
    reduction and a host sync the first time a tensor is seen."""
    global _forced_path
    if path not in FORWARD_PATHS:
        raise ValueError(f'forward path must be one of {sorted(FORWARD_PATHS)}; got {path!r}')
    _forced_path = path
    _lib.check(_lib.load().torbi_hip_set_forward_path(FORWARD_PATHS[path]),
               'torbi_hip_set_forward_path')


def scan_stats(workspace: torch.Tensor, batch: int, frames: int, states: int,
               path: str = 'pruned') -> Optional[torch.Tensor]:
    """Enqueue a copy of the scan statistics the last time-resident (or held-matrix) decode left in
    `workspace` (include/torbi_hip.h, torbi_hip_scan_stats): a (128,) int32 device tensor, valid once the
    current stream reaches it; None when the shape takes neither path."""
    lib = _lib.load()
    out = torch.empty((128,), dtype=torch.int32, device=workspace.device)
    rc = lib.torbi_hip_scan_stats(workspace.data_ptr(), workspace.numel(), batch, frames, states, out.data_ptr(),
                                  workspace.device.index or 0,
                                  ctypes.c_void_p(torch.cuda.current_stream(workspace.device).cuda_stream),
                                  _path_flag(path))
    if rc == -5:
        return None
    _lib.check(rc, 'torbi_hip_scan_stats')
    return out


def critical_blocks(stats: torch.Tensor) -> float:
    """Mean number of 16-entry list blocks on the critical path of a launch (host sync)."""
    host = (stats if not stats.is_cuda else stats.cpu()).to(torch.int64)
    # ([120], [121] hold clock ticks and [127] the workgroups that gave up waiting: not counts of passes)
    return float(host[:64].sum()) / max(1.0, float(host[64:120].sum()))


def delivered_clock_hz(stats: torch.Tensor) -> Optional[float]:
    """Shader clock the forward kernel of the last decode ran at, from the ticks its workgroup 0 left in the statistics
    (include/torbi_hip.h, torbi_hip_scan_stats [120], [121]: shader-clock ticks over 100 MHz wall-clock ticks); None for a
    route that leaves none (host sync)."""
    host = (stats if not stats.is_cuda else stats.cpu()).to(torch.int64)
    ticks, wall = int(host[120]) & 0xffffffff, int(host[121]) & 0xffffffff
    return ticks / wall * 1e8 if ticks > 0 and wall > 0 else None


def _choose_path(trans: torch.Tensor, original: torch.Tensor, batch: int, states: int) -> str:
    """AUTO's look at the transition matrix: 'dense' for narrow bands, 'pruned' otherwise; 'auto' (the library's
    own fallbacks) for shapes where the value-only paths offer no choice."""
    if batch < 32 or states < 64 or states > 4096:   # dense needs B >= 32
        return 'auto'
    kept = state.notes(original)                     # None for tensors without a version counter: looked at, not kept
    reach = kept.get(('reach', states)) if kept is not None else None
    if reach is None and trans.is_cuda:
        # ONE blocking look per tensor version on the device, shared with the band kernel's (torbi_hip_band_reach: a small
        # kernel and a sync, ~30 us): the widest row's reach instead of the mean of the rows' -- no second reduction, no
        # second sync.  (Routing the first call blind instead would cost a decode on the wrong kernel: milliseconds.)
        left, right = _device_reach(trans, original, states)
        reach = min(1.0, (left + right + 1) / states) if left >= 0 else 0.0       # (a matrix without a finite entry: 0)
        if kept is not None:
            kept[('reach', states)] = reach
    if reach is None:
        finite = trans != float('-inf')
        index = torch.arange(states, device=trans.device)
        lo = torch.where(finite, index, states).amin(dim=1)
        hi = torch.where(finite, index, -1).amax(dim=1)
        reach = float((hi - lo + 1).clamp(min=0).float().mean().item()) / states
        if kept is not None:
            kept[('reach', states)] = reach
    return 'dense' if 0.0 < reach < BANDED_RANGE else 'pruned'


ROUTES = {0: 'generic', 1: 'dense', 3: 'resident', 4: 'rows', 5: 'cluster', 6: 'held', 7: 'small', 8: 'band'}      # (2: retired in round 4)


def forward_path(batch: int, states: int, path: Optional[str] = None, device: int = 0) -> str:
    """Which forward recurrence the library runs for one (batch, states) problem under `path` (None = the process
    default): 'generic', 'dense', 'resident', 'cluster', 'held', 'rows' (the pruned recurrence for batches of <= 16 items)
    or 'small' (up to 64 states: one wavefront per sequence)."""
    code = _lib.load().torbi_hip_forward_path_on(int(batch), int(states), int(device),
                                                 _path_flag(_forced_path if path is None else path))
    if code < 0:
        _lib.check(code, 'torbi_hip_forward_path_on')
    return ROUTES[code]


def last_forward_kernel() -> str:
    """Name of the forward kernel this thread's most recent decode launched, as rocprofv3 spells it
    (include/torbi_hip.h, torbi_hip_last_forward_kernel); '' before the first decode."""
    buffer = ctypes.create_string_buffer(192)
    _lib.check(_lib.load().torbi_hip_last_forward_kernel(buffer, len(buffer)), 'torbi_hip_last_forward_kernel')
    return buffer.value.decode()


def uniform_supported(states: int) -> bool:
    """Shapes the uniform-transition entry point covers (include/torbi_hip.h)."""
    return states % 4 == 0 and states <= 4096


def decode_uniform(
    observation: torch.Tensor,
    batch_frames: torch.Tensor,
    log_transition: float,
    initial: torch.Tensor,
    probabilities: bool = False,
) -> torch.Tensor:
    """`decode` for a transition matrix whose entries all equal `log_transition`

    The reference's default (`from_probabilities` without a transition builds
    `torch.full((S, S), log(1/S))`, torbi/core.py:175-180).  O(S) per timestep, no scratch,
    HBM-bound; bit-identical to `decode` on the materialised matrix.  Falls back to
    materialising the matrix for shapes `uniform_supported` rejects.

    probabilities: `observation` holds probabilities (the default input of `from_probabilities`); the reference's
        `torch.log` and epsilon round trip `log(exp(x) + tiny)` (torbi/core.py:189-197) are applied to every element as
        it is read (include/torbi_hip.h, torbi_hip_viterbi_decode_uniform_probabilities) -- one pass over the
        observations for the whole default call; `observation` is not written.  `initial` stays in log space.
    """
    B, T, S = observation.shape
    if probabilities and not (uniform_supported(S) and observation.dtype == torch.float32):
        tiny = torch.finfo(torch.float32).tiny             # (the steps of torbi/core.py:189-197, one by one)
        scores = torch.log(observation).to(dtype=torch.float32)
        scores.exp_()
        scores += tiny
        scores.log_()
        return decode_uniform(scores, batch_frames, log_transition, initial)
    if not uniform_supported(S):
        transition = torch.full((S, S), float(log_transition), dtype=torch.float32,
                                device=observation.device)
        return decode(observation, batch_frames, transition, initial)
    _check_inputs(observation, batch_frames, torch.empty((S, S), dtype=torch.float32, device='meta'),
                  initial)
    _require_gpu()
    lib = _lib.load()
    home = observation.device
    device = home if home.type == 'cuda' else torch.device('cuda', torch.cuda.current_device())
    obs = observation.to(device).contiguous()
    frames = batch_frames.to(device).contiguous()
    init = initial.to(device).contiguous()
    indices = torch.empty((B, T), dtype=torch.int32, device=device)
    if B == 0:
        return indices.to(home)
    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(device).cuda_stream
    entry = lib.torbi_hip_viterbi_decode_uniform_probabilities if probabilities else lib.torbi_hip_viterbi_decode_uniform
    code = entry(obs.data_ptr(), frames.data_ptr(), float(log_transition), init.data_ptr(),
                 indices.data_ptr(), B, T, S, index, ctypes.c_void_p(stream))
    if code == -5:     # TORBI_HIP_EUNSUPPORTED (e.g. a misaligned view): materialise
        if probabilities:
            tiny = torch.finfo(torch.float32).tiny
            obs = torch.log(obs)
            obs.exp_()
            obs += tiny
            obs.log_()
        transition = torch.full((S, S), float(log_transition), dtype=torch.float32, device=device)
        return decode(obs, batch_frames, transition, initial).to(home)      # (like every other return of this function)
    _lib.check(code, 'torbi_hip_viterbi_decode_uniform')
    return indices if home == device else indices.to(home)


def read_posterior(workspace, batch_frames, batch, frames, states, path: Optional[str] = None):
    """Final posterior rows (N, S) of the last decode that used `workspace` (diagnostic); `path` as given to
    that decode."""
    lib = _lib.load()
    device = workspace.device
    out = torch.empty((batch, states), dtype=torch.float32, device=device)
    bf = batch_frames.to(device=device, dtype=torch.int32).contiguous()
    stream = torch.cuda.current_stream(device).cuda_stream
    _lib.check(lib.torbi_hip_read_posterior(
        workspace.data_ptr(), workspace.numel(), bf.data_ptr(), out.data_ptr(),
        batch, frames, states, device.index or 0, ctypes.c_void_p(stream),
        _path_flag(_forced_path if path is None else path)),
        'torbi_hip_read_posterior')
    return out


def epsilon_clamp_(x: torch.Tensor) -> torch.Tensor:
    """In place `x <- log(exp(x) + tiny)`: the three elementwise passes of reference
    torbi/core.py:193-197 (`exp_`, `+= tiny`, `log_`) fused into one (bit-identical to the torch
    ops on the same device; tested).  Falls back to the torch ops for tensors the fused kernel
    does not take (non-CUDA, non-fp32, non-contiguous)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.data_ptr() % 16 == 0):
        torch.exp_(x)
        x += torch.finfo(torch.float32).tiny
        torch.log_(x)
        return x
    lib = _lib.load()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _lib.check(lib.torbi_hip_epsilon_clamp(x.data_ptr(), x.numel(), x.device.index or 0,
                                           ctypes.c_void_p(stream)), 'torbi_hip_epsilon_clamp')
    return x


def log_epsilon_clamp(probabilities: torch.Tensor) -> Optional[torch.Tensor]:
    """`log(exp(log(p)) + tiny)` of a float32 HIP tensor in ONE pass, out of place: upstream's `torch.log(observation)`
    (torbi/core.py:189-191) followed by the epsilon round trip (core.py:193-197), bit-identical to the four torch ops
    on the same device (tested).  None for tensors the fused kernel does not take (the caller runs the torch ops)."""
    p = probabilities
    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() % 16 == 0):
        return None
    out = torch.empty_like(p)
    if out.data_ptr() % 16:
        return None
    stream = torch.cuda.current_stream(p.device).cuda_stream
    _lib.check(_lib.load().torbi_hip_log_epsilon_clamp(p.data_ptr(), out.data_ptr(), p.numel(), p.device.index or 0,
                                                       ctypes.c_void_p(stream)), 'torbi_hip_log_epsilon_clamp')
    return out


def fill_synthetic(shape, stream_id, seed=0, device=None, start=0):
    """Device-side torbi_amd.synth.scores(): deterministic fp32 scores in (-16, 0]."""
    _require_gpu()
    lib = _lib.load()
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.empty(shape, dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    _lib.check(lib.torbi_hip_fill_synthetic(
        out.data_ptr(), out.numel(), start, stream_id, seed, device.index or 0,
        ctypes.c_void_p(stream)), 'torbi_hip_fill_synthetic')
    return out
