"""Tensor and file API: the host-side mirror of reference torbi/core.py:110-473.

Same function names, argument order, defaults and return conventions as the reference; the
decode itself runs on an MI355X through `torbi_amd.decode` (C ABI, include/torbi_hip.h).
Differences that follow from "GPU only" are stated in each docstring and in INTEGRATION.md.
"""
import math
import os
from typing import Dict, List, Optional, Union

import torch

from . import data as _data
from . import timer
from .viterbi import decode, decode_uniform, epsilon_clamp_, log_epsilon_clamp, uniform_supported

# reference torbi/config/defaults.py:80,83
BATCH_SIZE = 512
NUM_WORKERS = 0
# reference torbi/config/defaults.py:41,44: chunked decoding of long sequences (torbi_amd/chunk.py); off when None
MIN_CHUNK_SIZE = None
ENTROPY_THRESHOLD = 0.5
# batches of a many-file job decoded per launch group (torbi_amd.DecodePipeline / decode_batches): 8 x 512 items
# give every compute unit of an MI355X one 16-item workgroup of the time-resident kernel
GROUP_SIZE = 8
# from_files_to_files reads plain float32 observation files straight into pinned batch buffers (torbi_amd/fastio.py)
# instead of torch.load + collate in DataLoader workers; False = the reference's loader for every file
DIRECT_FILE_IO = os.environ.get('TORBI_DIRECT_FILE_IO', '1') != '0'
# threads that write the per-file outputs while the next launch group is decoded (0 = save on the calling thread)
SAVE_THREADS = 2
# A many-file job's scratch -- the launch-group pipeline's workspaces (2 x GROUP_SIZE buffers of 1.5-2.7 GB each at 1440
# states) and the pinned / device staging slabs -- is FREED WITH THE JOB, like the reference's allocations.  A process that
# runs job after job can keep it (TORBI_KEEP_JOB_MEMORY=1, or set this to True): the second job then allocates nothing
# (20 000 files: 1.3 s against 4.4 s for the first call) at the price of those gigabytes staying out of torch's caching
# allocator until `release_job_memory()`.
KEEP_JOB_MEMORY = os.environ.get('TORBI_KEEP_JOB_MEMORY', '0') != '0'
# The file batches of a GPU job cross the host link through a ring of RING_CHUNKS pinned chunks of RING_CHUNK_BYTES (powers
# of two: the pinned allocator rounds up to one), not through whole-batch pinned slabs; 0 chunks = whole-batch slabs.
RING_CHUNKS = int(os.environ.get('TORBI_RING_CHUNKS', '4'))
# probability files (log_probs=False, the reference's default) take the staged route too: log() + epsilon round trip as one
# pass in place in the device slab; False = pinned batch -> from_probabilities' own device move, log and clamp (rounds 2-5)
STAGE_PROBABILITIES = os.environ.get('TORBI_STAGE_PROBABILITIES', '1') != '0'
RING_CHUNK_BYTES = int(os.environ.get('TORBI_RING_CHUNK_MB', '256')) << 20


def _compute_device(gpu):
    if gpu == 'mps':
        raise RuntimeError('the MPS backend of the reference is out of scope on MI355X')
    if not torch.cuda.is_available():
        raise RuntimeError(
            f'from_probabilities(gpu={gpu!r}) needs a HIP device and PyTorch-ROCm reports none; there is no CPU '
            'fallback for a GPU request (gpu=None selects the CPU operator, like upstream)')
    return torch.device(f'cuda:{gpu}')


def _from_probabilities_cpu(observation, batch_frames, transition, initial, log_probs, num_threads):
    """`gpu=None`: the reference's CPU route (torbi/core.py:145-201 with device = 'cpu'), decoded by the host twin of
    the operator (include/torbi_cpu.h).  Same steps in the same order, materialised uniform defaults included."""
    from .viterbi import decode_cpu
    batch, frames, states = observation.shape
    device = torch.device('cpu')
    tiny = torch.finfo(torch.float32).tiny
    if batch_frames is None:
        batch_frames = torch.full((batch,), frames, dtype=torch.int32, device=device)
    batch_frames = batch_frames.to(dtype=torch.int32, device=device)
    if initial is None:
        initial = torch.full((states,), math.log((1. / states) + tiny), dtype=torch.float32, device=device)
    else:
        if not log_probs:
            initial = torch.log(initial)
        initial = initial.to(device)
    if transition is None:
        transition = torch.full((states, states), math.log(1. / states), dtype=torch.float32, device=device)
    else:
        if not log_probs:
            transition = torch.log(transition)
        transition = transition.to(device)
    if not log_probs:
        observation = torch.log(observation)
    observation = observation.to(device=device, dtype=torch.float32)
    torch.exp_(observation)
    observation += tiny
    torch.log_(observation)
    with timer.context('torbi'):
        return decode_cpu(observation, batch_frames, transition, initial, num_threads=num_threads)


def _prepared_transition(transition: torch.Tensor, log_probs: bool, device) -> torch.Tensor:
    """log() (unless `log_probs`) and device move of the transition matrix (core.py:181-187), remembered with the
    caller's tensor (object and version, torbi_amd/state.py): repeated calls with one matrix then hand torbi_amd.decode
    the SAME device tensor, which is what its structure look and path measurements hang off."""
    from . import state
    kept = state.notes(transition)             # None under torch.inference_mode(): nothing to remember it by
    key = ('prepared', bool(log_probs), str(device))
    if kept is not None and key in kept:
        return kept[key]
    prepared = (transition if log_probs else torch.log(transition)).to(device)
    if kept is not None:
        kept[key] = prepared
    return prepared


# batches of host probabilities from this size on take their log() into a pooled pinned buffer (_host_log)
HOST_LOG_POOL_BYTES = 32 << 20


def _host_log(observation: torch.Tensor) -> torch.Tensor:
    """`torch.log(observation)` where the observation lives, like upstream (core.py:189-191: the log is taken BEFORE the device
    move, so a host batch is logged by the host and the operator sees the host's roundings).  A large float32 host batch is
    logged into a pinned buffer of the process-wide pool (torbi_amd/slabs.py; `release_job_memory()` frees it): the same
    kernel, the same bits, but no fresh 1.5 GB of first-touched pages per call and an asynchronous copy at the host link's rate
    behind it -- 512 x 500 x 1440: 398 -> ~65 ms per call."""
    nbytes = observation.numel() * 4
    if (observation.device.type != 'cpu' or observation.dtype != torch.float32 or nbytes < HOST_LOG_POOL_BYTES
            or observation.requires_grad or not torch.cuda.is_available()):          # (`out=` is not for tensors in a graph)
        return torch.log(observation)
    from . import slabs
    slab = slabs.pool(None).take(nbytes, limit=2)
    out = slab[:nbytes].view(torch.float32).view(observation.shape)
    torch.log(observation, out=out)
    out.torbi_slab = slab
    return out


def from_probabilities(
    observation: torch.Tensor,
    batch_frames: Optional[torch.Tensor] = None,
    transition: Optional[torch.Tensor] = None,
    initial: Optional[torch.Tensor] = None,
    log_probs: bool = False,
    gpu: Optional[int] = None,
    num_threads: Optional[int] = 1,
    _pipeline=None,
    _model: Optional[dict] = None,
    _prepared: bool = False
) -> torch.Tensor:
    """Viterbi-decode a batch of per-frame state distributions on HIP device `gpu` (or on the CPU: `gpu=None`).

    Same steps, in the same order, as reference torbi/core.py:110-208: `batch_frames` defaults to
    every frame (int32); a missing `initial` becomes log(1/S + tiny) and a missing `transition`
    log(1/S); inputs are taken through `log()` unless `log_probs`; the observation is cast to
    fp32 on the compute device and goes through the in-place round trip `log(exp(x) + tiny)`;
    then `decode`.

    Args:
        observation: (batch, frames, states) scores of every state at every frame
        batch_frames: (batch,) valid frames per item; None = all of them
        transition: (states, states) matrix indexed [next, prev]; None = uniform (decoded by the
            O(states)-per-frame kernel, same indices)
        initial: (states,) distribution over the first frame's states; None = uniform
        log_probs: the tensors are already natural-log probabilities
        gpu: HIP device index.  None selects the CPU operator like upstream (torbi/core.py:147-150): its host twin
            (`decode_cpu`, include/torbi_cpu.h), same indices.  A GPU request without a device raises
        num_threads: worker threads of the CPU operator (upstream: torch's global thread count,
            torbi/viterbi.py:51-52); ignored on the GPU

    Returns:
        (batch, frames) int32 indices of the most likely state sequence of every item
    """
    if gpu is None:
        return _from_probabilities_cpu(observation, batch_frames, transition, initial, log_probs, num_threads)
    batch, frames, states = observation.shape
    device = _compute_device(gpu)
    tiny = torch.finfo(torch.float32).tiny

    if batch_frames is None:
        batch_frames = torch.full((batch,), frames, dtype=torch.int32, device=device)
    batch_frames = batch_frames.to(dtype=torch.int32, device=device)

    # `_model` (from_dataloader): initial/transition are the same objects for every batch, so their
    # log(), device move and (in torbi_amd.decode) structure look are done once, not per batch
    if _model is not None and 'initial' in _model:
        initial, transition, uniform = _model['initial'], _model['transition'], _model['uniform']
    else:
        # Default to uniform initial probabilities (core.py:161-166)
        if initial is None:
            initial = torch.full(
                (states,), math.log((1. / states) + tiny), dtype=torch.float32, device=device)
        else:
            if not log_probs:
                initial = torch.log(initial)
            initial = initial.to(device)

        # Default to uniform transition probabilities (core.py:175-180).  The reference
        # materialises torch.full((S, S), log(1/S)); a matrix of identical entries is decoded by the
        # O(S)-per-timestep entry point instead, with identical results (decode_uniform).
        uniform = None
        if transition is None:
            uniform = float(torch.tensor(math.log(1. / states), dtype=torch.float32))
        else:
            transition = _prepared_transition(transition, log_probs, device)
        if _model is not None:
            _model.update(initial=initial, transition=transition, uniform=uniform)

    # The whole default call -- probabilities in, no transition given -- as ONE pass over the observations: log(), the
    # epsilon round trip and the O(S)-per-frame decode in one kernel (same values, same indices; csrc/uniform_decode.hpp)
    if (uniform is not None and not log_probs and observation.device == device
            and observation.dtype == torch.float32 and uniform_supported(states)):
        with timer.context('torbi'):
            return decode_uniform(observation, batch_frames, uniform, initial, probabilities=True)

    # Ensure observation probabilities are in log space (core.py:189-191).  Probabilities that already live on the
    # compute device go through log() and the epsilon round trip below in one pass (same values, tested bitwise)
    # (`_prepared`, the many-file driver: the batch is on the device, float32, and has been through both steps already --
    # in place in its staging slab, _Staging.decode)
    clamped = observation if _prepared else None
    if not log_probs and not _prepared:
        if observation.device == device:
            clamped = log_epsilon_clamp(observation)
        if clamped is None:
            observation = _host_log(observation)
    # non_blocking: a pinned host batch (data.loader) is copied asynchronously, so the copy of batch k+1
    # runs under the decode of batch k; pageable sources fall back to the synchronous path by themselves
    on_host = observation
    observation = observation.to(device=device, dtype=torch.float32, non_blocking=True)
    if getattr(on_host, 'torbi_slab', None) is not None:        # (_host_log's pooled buffer: free again once the copy has left)
        from . import slabs
        left = torch.cuda.Event()
        left.record(torch.cuda.current_stream(device))
        slabs.pool(None).give(on_host.torbi_slab, left)

    # Add epsilon for stability (core.py:193-197; in place, like the reference): exp_, += tiny,
    # log_ as ONE elementwise pass on the device
    if clamped is not None:
        observation = clamped
    else:
        epsilon_clamp_(observation)

    # Decode, inside the same timing scope as upstream's (core.py:200-206; torbi_amd/timer.py)
    with timer.context('torbi'):
        if uniform is not None:
            indices = decode_uniform(observation, batch_frames, uniform, initial)
        elif _pipeline is not None:
            # asynchronous: valid after _pipeline.wait(indices); the caller owns the host copy
            return _pipeline.decode(observation, batch_frames, transition, initial)
        else:
            indices = decode(observation, batch_frames, transition, initial, num_threads=num_threads)
    return indices


def from_file(
    input_file: Union[str, os.PathLike],
    transition_file: Optional[Union[str, os.PathLike]] = None,
    initial_file: Optional[Union[str, os.PathLike]] = None,
    log_probs: bool = False,
    gpu: Optional[int] = None,
    num_threads: Optional[int] = 1
) -> torch.Tensor:
    """`from_probabilities` for one `torch.save`d (frames, states) observation; returns
    (1, frames) indices (reference core.py:211-267).  `transition_file` / `initial_file` hold a
    (states, states) / (states,) tensor and default to uniform."""
    transition, initial = _load_model(transition_file, initial_file, log_probs, clamp=False)
    return from_probabilities(torch.load(input_file)[None], None, transition, initial, log_probs, gpu,
                              num_threads)


def _load_model(transition_file, initial_file, log_probs, clamp):
    """The optional model files of the file entry points.  Upstream takes log() of a loaded
    transition exactly when `log_probs` is set, because from_probabilities will not (the files
    hold probabilities): plainly in from_file (core.py:246-247), with `+ tiny` in
    from_files_to_files (core.py:341-347).  Kept as is."""
    transition = None
    if transition_file:
        transition = torch.load(transition_file)
        if log_probs:
            transition = torch.log(transition + torch.finfo(transition.dtype).tiny if clamp else transition)
    return transition, (torch.load(initial_file) if initial_file else None)


def from_file_to_file(
    input_file: Union[str, os.PathLike],
    output_file: Union[str, os.PathLike],
    transition_file: Optional[Union[str, os.PathLike]] = None,
    initial_file: Optional[Union[str, os.PathLike]] = None,
    log_probs: bool = False,
    gpu: Optional[int] = None,
    num_threads: Optional[int] = None
) -> None:
    """`from_file`, with the (1, frames) indices written to `output_file` (core.py:270-307)."""
    torch.save(from_file(input_file, transition_file, initial_file, log_probs, gpu, num_threads), output_file)


def from_files_to_files(
    input_files: List[Union[str, os.PathLike]],
    output_files: List[Union[str, os.PathLike]],
    transition_file: Optional[Union[str, os.PathLike]] = None,
    initial_file: Optional[Union[str, os.PathLike]] = None,
    log_probs: bool = False,
    gpu: Optional[int] = None,
    num_threads: Optional[int] = None,
    lengths: Optional[List[int]] = None,
    num_workers: Optional[int] = None
) -> None:
    """Decode many observation files, one index file each (core.py:310-368).

    Files are batched `BATCH_SIZE` (512) at a time in the given order, zero-padded to the
    longest item of the batch (reference torbi/data/collate.py:24-33) and each output holds
    the first `frames` indices of its item (core.py:449-457).

    Two additions over the reference signature, both result-neutral (items are independent
    and every output is written to the file mapped to its input):
        lengths      frames per input file, when known: batches are then formed from files
                     of similar length (longest first), which removes most of the padding a
                     ragged collection costs (every padded frame is a full recurrence step)
        num_workers  DataLoader workers for torch.load (reference default 0, loader.py:19-25); reader threads
                     of the direct file reader (below)

    Plain float32 files are not taken through `torch.load` + `collate` at all: their payload is read from the
    `torch.save` container straight into its row of a pinned batch buffer (torbi_amd/fastio.py; same tuples, same
    zero padding), and outputs are written by `SAVE_THREADS` threads while the next batches are decoded.  Files
    the direct reader does not take (other dtypes or layouts, chunked decoding) go the reference's way.
    """
    transition, initial = _load_model(transition_file, initial_file, log_probs, clamp=True)
    mapping = dict(zip(input_files, output_files))

    if lengths is not None:
        if len(lengths) != len(input_files):
            raise ValueError('lengths must have one entry per input file')
        order = sorted(range(len(input_files)), key=lambda k: (-int(lengths[k]), k))
        input_files = [input_files[k] for k in order]

    batches = None
    if DIRECT_FILE_IO:
        from . import fastio
        batches = fastio.open_batches(input_files, BATCH_SIZE, threads=num_workers,
                                      pin_memory=gpu is not None and torch.cuda.is_available(), gpu=gpu is not None)
    from_dataloader(
        dataloader=batches if batches is not None else _data.loader(input_files, num_workers=num_workers),
        output_files=mapping,
        transition=transition,
        initial=initial,
        log_probs=log_probs,
        gpu=gpu,
        num_threads=num_threads)


def from_dataloader(
    dataloader: torch.utils.data.DataLoader,
    output_files: Dict[
        Union[str, bytes, os.PathLike],
        Union[str, bytes, os.PathLike]],
    transition: Optional[torch.Tensor] = None,
    initial: Optional[torch.Tensor] = None,
    log_probs: bool = False,
    gpu: Optional[int] = None,
    num_threads: Optional[int] = 1
) -> None:
    """Decode every batch a `data.loader` yields and save each item under `output_files[input]`
    (core.py:376-463).

    The reference loop is serial (decode, copy back, save, next batch).  Here `GROUP_SIZE` consecutive
    batches are decoded together (torbi_amd.DecodePipeline -> decode_batches: one time-resident forward
    launch for the group when it has enough items), groups alternate between two HIP streams, and a
    batch's indices are copied back and saved while the following group is collected and decoded.
    Outputs are identical.
    """
    from .pipeline import DecodePipeline
    import collections
    from concurrent.futures import ThreadPoolExecutor
    pipe, give_back = None, None
    savers = ThreadPoolExecutor(max_workers=SAVE_THREADS) if SAVE_THREADS > 0 else None
    written = collections.deque()

    def write(function, *args):
        if savers is None:
            function(*args)
            return
        written.append(savers.submit(function, *args))
        while len(written) > 4096:            # bound the queue; surfaces a failed save early
            written.popleft().result()

    if DIRECT_FILE_IO:
        # same file contents as save / save_masked (core.py:466-473), written from a prebuilt container image
        import functools
        from .fastio import save_indices, save_index_rows
        save_rows = functools.partial(save_index_rows, gpu=gpu is not None)

        def store(tensor, file, length):
            save_indices(tensor if length is None else tensor[..., :length], file)
    else:
        def store(tensor, file, length):
            save(tensor, file) if length is None else save_masked(tensor, file, length)

    stage = None
    if gpu is not None and torch.cuda.is_available():
        pipe, give_back = _job_pipeline(torch.device('cuda', gpu))
        stage = _Staging(torch.device('cuda', gpu))
        if (log_probs or STAGE_PROBABILITIES) and hasattr(dataloader, 'stage') and getattr(dataloader, 'pin_memory', False):
            dataloader.stage = stage.upload           # (fastio.FileBatches: copies start in the assembling threads)
            if RING_CHUNKS > 0 and hasattr(dataloader, 'stage_rows'):
                dataloader.stage_rows = stage.upload_rows        # ... chunk by chunk through a small pinned ring
            if hasattr(dataloader, 'start'):
                dataloader.start()                    # the first batches are read while the transition matrix is prepared
        if transition is not None:
            # the one look at the transition matrix that costs a host sync (torbi_amd.viterbi._choose_path) happens now,
            # while the device is idle, not at the first launch group with several batches' copies queued behind it
            from .viterbi import _choose_path
            prepared = _prepared_transition(transition, log_probs, torch.device('cuda', gpu))
            _choose_path(prepared, prepared, BATCH_SIZE, prepared.shape[-1])

    def finish(item):
        indices, input_filenames, batch_frames, batch_chunks = item
        if pipe is not None:
            pipe.wait(indices)
        filenames = [output_files[file] for file in input_filenames]
        rows = indices.cpu().detach()
        if any(int(count) != 1 for count in batch_chunks):
            # files that were cut into pieces (torbi_amd/chunk.py): join each file's rows again (core.py:438-448)
            for joined, filename in zip(_data.separate(rows, batch_chunks, batch_frames.cpu()), filenames):
                write(store, joined, filename, None)
        elif DIRECT_FILE_IO:
            write(save_rows, rows, filenames, batch_frames.cpu().tolist())       # one task: the whole batch, native threads
        else:
            for row, filename, frames in zip(rows, filenames, batch_frames.cpu().tolist()):
                write(store, row, filename, frames)

    # batches stay outstanding until a whole group behind them has been enqueued: the group being collected,
    # the group being decoded and the batch being saved overlap
    outstanding = collections.deque()
    keep = 1 if pipe is None else pipe.group * (pipe.depth - 1) + 1
    model = {}
    starved = getattr(dataloader, 'more_ready', None)
    try:
        for observation, batch_frames, batch_chunks, input_filenames in dataloader:
            if stage is not None and (log_probs or STAGE_PROBABILITIES) and stage.takes(observation):
                # a float32 batch in pinned memory (or already on its way): copy on the copy stream into a pooled device
                # slab, log() unless `log_probs` + epsilon round trip (core.py:189-197) on the preparation stream, decode
                # on the pipeline's streams
                indices = stage.decode(observation, batch_frames, transition, initial, gpu, num_threads, pipe, model, log_probs)
            else:
                indices = from_probabilities(
                    observation=observation,
                    batch_frames=batch_frames,
                    transition=transition,
                    initial=initial,
                    log_probs=log_probs,
                    gpu=gpu,
                    num_threads=num_threads,
                    _pipeline=pipe,
                    _model=model)
            outstanding.append((indices, input_filenames, batch_frames, batch_chunks))
            if pipe is not None and starved is not None and not starved():
                pipe.flush()      # the reader is the slower side: decode what has arrived instead of waiting for a full group
            elif pipe is not None and stage is not None:
                pipe.flush_if_idle()      # ... and so is the host link: an idle device decodes what is ready
            while len(outstanding) > keep:
                finish(outstanding.popleft())
        while outstanding:
            finish(outstanding.popleft())
        while written:
            written.popleft().result()
    finally:
        if savers is not None:
            savers.shutdown(wait=True)
        if give_back is not None:
            if pipe is not None:
                pipe.synchronize()         # (an exception above may have left batches collected or in flight)
            give_back()


_job_pipelines = {}           # device -> [DecodePipeline kept between many-file jobs, lock held by the job using it]
_job_pipelines_lock = __import__('threading').Lock()     # (jobs may be started from several host threads)


def _job_pipeline(device):
    """The launch-group pipeline of a many-file job on `device` and what gives it back.  With KEEP_JOB_MEMORY the
    pipeline (and with it its scratch) is created once per process and device, so that a second job finds everything
    allocated; a job that finds it taken -- another host thread is decoding files on the same device -- works with one
    of its own.  Without (the default) every job has its own pipeline, dropped -- with the staging slabs -- when the job
    ends.  `release_job_memory()` drops what is kept."""
    import threading
    from .pipeline import DecodePipeline
    from ._lib import MAX_BATCHES
    if not KEEP_JOB_MEMORY:
        return DecodePipeline(device, depth=2, group=GROUP_SIZE), release_job_memory
    key = str(device)
    with _job_pipelines_lock:
        kept = _job_pipelines.get(key)
        if kept is None or kept[0].group != max(1, min(int(GROUP_SIZE), MAX_BATCHES)):     # (GROUP_SIZE changed)
            kept = _job_pipelines[key] = [DecodePipeline(device, depth=2, group=GROUP_SIZE), threading.Lock()]
    if kept[1].acquire(blocking=False):
        return kept[0], kept[1].release
    return DecodePipeline(device, depth=2, group=GROUP_SIZE), lambda: None


def release_job_memory() -> None:
    """Free what the many-file jobs keep between calls (KEEP_JOB_MEMORY): the pipelines' scratch and the staging slabs
    that no job is using."""
    from . import slabs
    with _job_pipelines_lock:
        _job_pipelines.clear()
    slabs.release()


class _Staging:
    """Host-to-device staging of the many-file job's batches: a COPY stream that carries nothing but the H2D copies (so a
    copy never waits behind a kernel that waits for a compute unit -- a launch group's forward kernel holds every one
    of them for tens of milliseconds), a PREPARATION stream for the epsilon round trip, pooled buffers on both sides
    (torbi_amd/slabs.py).  Same operations on the same values as from_probabilities (core.py:189-197)."""

    def __init__(self, device):
        from . import slabs
        self.device = device
        self.pool = slabs.pool(device)
        self.host_pool = slabs.pool(None)
        streams = _staging_streams.get(str(device))
        if streams is None:
            streams = _staging_streams[str(device)] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        self.copy, self.prep = streams

    @staticmethod
    def takes(observation) -> bool:
        return hasattr(observation, 'torbi_copied') or (
            observation.device.type == 'cpu' and observation.dtype == torch.float32 and observation.is_contiguous()
            and observation.is_pinned())

    def upload(self, observation, batch_frames):
        """Start the host-to-device copy of a pinned batch (any thread: the reader's assembling threads call this as soon
        as a batch exists): the batch on a pooled device slab, its lengths as int32, the event behind both."""
        nbytes = observation.numel() * 4
        slab = self.pool.take(nbytes)
        staged = slab[:nbytes].view(torch.float32).view(observation.shape)
        # (the lengths go first and from pinned memory: a pageable source would make the copy synchronous, and issued
        # behind the batch it would hold the calling thread until the whole batch has crossed the link)
        lengths = batch_frames.to(torch.int32).pin_memory()
        with torch.cuda.stream(self.copy):
            staged.torbi_lengths = lengths.to(self.device, non_blocking=True)
            staged.copy_(observation, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.copy)
        host_slab = getattr(observation, 'torbi_slab', None)
        if host_slab is not None:
            self.host_pool.give(host_slab, copied)       # the reader may refill it once the copy has left
        staged.torbi_copied = copied
        staged.torbi_device_slab = slab
        staged.torbi_keep = (observation, lengths)       # (pinned sources stay alive until the copy has been issued)
        return staged

    def upload_rows(self, shape, batch_frames, fill):
        """`upload` for a batch that has not been read yet (fastio.FileBatches.stage_rows): `fill(address, first, k)` reads
        rows first .. first + k - 1 to `address`.  The rows pass through a ring of RING_CHUNKS pinned chunks of
        RING_CHUNK_BYTES, each copied to its place in the pooled device slab as soon as it has been read."""
        count, longest, states = shape
        row_bytes = 4 * longest * states
        nbytes = count * row_bytes
        slab = self.pool.take(nbytes)
        staged = slab[:nbytes].view(torch.float32).view(shape)
        lengths = batch_frames.to(torch.int32).pin_memory()
        with torch.cuda.stream(self.copy):
            staged.torbi_lengths = lengths.to(self.device, non_blocking=True)
        per = max(1, RING_CHUNK_BYTES // row_bytes)
        copied = None
        for first in range(0, count, per):
            k = min(per, count - first)
            chunk = self.host_pool.take(max(RING_CHUNK_BYTES, k * row_bytes), limit=RING_CHUNKS, exact=True)
            fill(chunk.data_ptr(), first, k)
            with torch.cuda.stream(self.copy):
                slab[first * row_bytes:(first + k) * row_bytes].copy_(chunk[:k * row_bytes], non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(self.copy)
            self.host_pool.give(chunk, copied)           # refilled once its copy has left
        staged.torbi_copied = copied
        staged.torbi_device_slab = slab
        staged.torbi_keep = (lengths,)
        return staged

    def decode(self, observation, batch_frames, transition, initial, gpu, num_threads, pipe, model, log_probs=True):
        staged = observation if hasattr(observation, 'torbi_copied') else self.upload(observation, batch_frames)
        slab, copied = staged.torbi_device_slab, staged.torbi_copied
        self.prep.wait_event(copied)
        with torch.cuda.stream(self.prep):
            # (from_probabilities: already on the device, float32; log-probabilities -> epsilon round trip in place;
            # probabilities -> log() and the round trip as ONE pass, in place in the slab -- the batch is this job's own
            # copy of the files -- instead of a fresh 1.5-2.7 GB tensor per batch; then the pipeline, whose side stream
            # waits for the readiness event recorded on the stream that is current here.  Probabilities with the default
            # uniform model go as they are: from_probabilities decodes them in one pass, log() included.)
            prepared = False
            if not log_probs and transition is not None and staged.data_ptr() % 16 == 0:
                from . import _lib
                import ctypes
                _lib.check(_lib.load().torbi_hip_log_epsilon_clamp(
                    staged.data_ptr(), staged.data_ptr(), staged.numel(), self.device.index or 0,
                    ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), 'torbi_hip_log_epsilon_clamp')
                prepared = True
            indices = from_probabilities(observation=staged, batch_frames=staged.torbi_lengths, transition=transition,
                                         initial=initial, log_probs=log_probs, gpu=gpu, num_threads=num_threads,
                                         _pipeline=pipe, _model=model, _prepared=prepared)
        if pipe is not None and model.get('uniform') is None:
            pipe.when_done(indices, lambda event, slab=slab: self.pool.give(slab, event))
        else:
            # decoded on the preparation stream itself (the uniform-transition entry, or no pipeline): nothing else
            # orders the consumer's stream -- `indices.cpu()` in from_dataloader runs on the current stream -- behind it
            done = torch.cuda.Event()
            done.record(self.prep)
            torch.cuda.current_stream(self.device).wait_event(done)
            self.pool.give(slab, done)
        return indices


_staging_streams = {}


def save(tensor, file):
    """Save tensor (core.py:466-468)"""
    torch.save(tensor.clone(), file)


def save_masked(tensor, file, length):
    """Save masked tensor (core.py:471-473)"""
    torch.save(tensor[..., :length].clone(), file)
