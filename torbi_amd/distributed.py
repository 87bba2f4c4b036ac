"""Single-node multi-GPU decoding: one process per GPU, batch items sharded across ranks.

The reference has no multi-device code (a single `gpu` index: torbi/core.py:147-150).  Batch
items are independent in every backend (serial `for b` at torbi/csrc/viterbi.cpp:65, one block
per item at torbi/csrc/cuda/viterbi.cu:58), so the batch axis shards with NO data-path
collective; time and state axes do not shard (a per-timestep exchange of the posterior would
dominate).  The only exchange is the optional gather of the decoded int32 indices
(<= 4*B*T bytes, 1 MB at B=512, T=500) -- `torch.distributed` all_gather, which is RCCL over
xGMI for the "nccl" backend on ROCm.  In the file-to-file flow each rank writes its own
outputs and only a barrier is needed.

Launch: `python -m torch.distributed.run --nproc-per-node N ...`; each rank binds
`cuda:LOCAL_RANK`.  The CPU tests run the same code over gloo with a stand-in decode function.
"""
import os
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def world(group=None):
    """(rank, world_size) of `group`, or (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* and bind the local GPU."""
    rank = int(os.environ.get('RANK', '0'))
    size = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    force = os.environ.get('TORBI_FORCE_DIST') == '1'    # exercise the RCCL path with one rank
    if (size > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kwargs = {}
        if backend == 'nccl':
            kwargs['device_id'] = torch.device('cuda', torch.cuda.current_device())
        dist.init_process_group(backend=backend, rank=rank, world_size=size, **kwargs)
    return rank, size, local


def shard_bounds(count: int, size: int, rank: int):
    """Contiguous block [lo, hi) of `count` items owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(count, size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_indices(local: torch.Tensor, count: int, group=None, force: Optional[bool] = None) -> torch.Tensor:
    """All-gather the per-rank (n_r, T) int32 index blocks into the full (count, T) tensor.

    Shards may differ by one row, so blocks are padded to the largest shard for the
    collective and trimmed afterwards.  With one rank there is nothing to gather and `local` is returned as it is,
    unless `force` (default: the environment's TORBI_FORCE_DIST=1, the switch `init_from_env` also reads) asks for the
    collective anyway -- that is how the RCCL branch is exercised on a single GPU.
    """
    rank, size = world(group)
    if force is None:
        force = os.environ.get('TORBI_FORCE_DIST') == '1'
    if size == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return local
    frames = local.shape[1]
    widest = (count + size - 1) // size
    padded = local
    if local.shape[0] < widest:
        padded = torch.zeros((widest, frames), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty((size * widest, frames), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    pieces = []
    for r in range(size):
        lo, hi = shard_bounds(count, size, r)
        pieces.append(out[r * widest:r * widest + (hi - lo)])
    return torch.cat(pieces, dim=0)


def decode_sharded(observation, batch_frames, transition, initial, group=None, gather=True,
                   decode_fn: Optional[Callable] = None, count: Optional[int] = None):
    """Decode this rank's contiguous block of a batch; optionally gather.

    Two ways to hand over the batch.  `count` None: `observation`/`batch_frames` are the FULL (B, T, S) /
    (B,) tensors on every rank (e.g. read from shared storage) and each rank slices its block.  `count` = B:
    they are ALREADY this rank's block `shard_bounds(B, size, rank)` -- nothing but the shard has to be
    resident on the rank (1.47 GB per 512 x 500 x 1440 batch otherwise).  transition/initial are replicated
    (8.3 MB at S=1440 -- each rank uploads its own copy, no broadcast needed).  Returns (B, T) indices on
    every rank when `gather`, else this rank's (n_r, T) block.
    """
    if decode_fn is None:
        from .viterbi import decode as decode_fn
    rank, size = world(group)
    if count is None:
        count = observation.shape[0]
        lo, hi = shard_bounds(count, size, rank)
        observation, batch_frames = observation[lo:hi].contiguous(), batch_frames[lo:hi].contiguous()
    else:
        lo, hi = shard_bounds(count, size, rank)
        if observation.shape[0] != hi - lo:
            raise ValueError(f'rank {rank} of {size} owns {hi - lo} of {count} items; got {observation.shape[0]}')
    local = decode_fn(observation, batch_frames, transition, initial)
    return gather_indices(local, count, group) if gather else local


def assign_batches(lengths: Sequence[int], batch_size: int, size: int) -> List[List[List[int]]]:
    """Split files (in the given order) into batches of `batch_size` like the reference's
    DataLoader (torbi/data/loader.py:19-25) and assign whole batches to ranks, longest
    padded cost first onto the least-loaded rank.  Returns per rank a list of batches, each a
    list of file positions.  Every file appears exactly once.
    """
    batches = [list(range(i, min(i + batch_size, len(lengths))))
               for i in range(0, len(lengths), batch_size)]
    cost = [max(lengths[i] for i in b) * len(b) for b in batches]   # padded frames decoded
    order = sorted(range(len(batches)), key=lambda k: (-cost[k], k))
    load = [0] * size
    plan = [[] for _ in range(size)]
    for k in order:
        r = min(range(size), key=lambda q: (load[q], q))
        plan[r].append(k)
        load[r] += cost[k]
    return [[batches[k] for k in sorted(ks)] for ks in plan]


def assign_files(lengths: Sequence[int], batch_size: int, size: int) -> List[List[int]]:
    """File positions per rank: the batches of `assign_batches`, flattened in order, a short trailing batch
    last (so re-batching a rank's list `batch_size` at a time reproduces whole batches)."""
    plan = assign_batches(lengths, batch_size, size)
    return [[i for batch in sorted(batches, key=lambda b: (len(b) < batch_size, b[0])) for i in batch]
            for batches in plan]


def from_files_to_files(input_files, output_files, transition_file=None, initial_file=None,
                        log_probs=False, gpu=None, num_threads=None, lengths=None, group=None,
                        decode_files: Optional[Callable] = None, num_workers: Optional[int] = None):
    """Multi-GPU form of torbi_amd.from_files_to_files: each rank decodes and saves the files of whole
    batches of its own in ONE call of the single-GPU entry point (so the model is loaded and prepared once
    per rank and consecutive batches share launch groups and streams); no collective besides the closing
    barrier.

    `lengths` (frames per file) lets batches be balanced by padded cost and is handed on, so every rank
    also forms its batches from files of similar length; without it batches are dealt round-robin.
    `gpu` defaults to this rank's current device.  Returns the number of files this rank decoded.
    """
    from . import core
    rank, size = world(group)
    if decode_files is None:
        decode_files = core.from_files_to_files
    if gpu is None and torch.cuda.is_available():
        gpu = torch.cuda.current_device()
    n = len(input_files)
    known = lengths is not None
    mine = assign_files(list(lengths) if known else [1] * n, core.BATCH_SIZE, size)[rank]
    if mine:
        extra = {'lengths': [lengths[i] for i in mine]} if known else {}
        if num_workers is not None:
            extra['num_workers'] = num_workers      # DataLoader workers for torch.load (reference default 0)
        decode_files([input_files[i] for i in mine], [output_files[i] for i in mine],
                     transition_file, initial_file, log_probs, gpu, num_threads, **extra)
    if size > 1:
        dist.barrier(group=group)
    return len(mine)
