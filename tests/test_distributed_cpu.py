"""World-size-2 gloo runs of the multi-GPU sharding logic on CPU.

The decode itself is replaced by an oracle-backed stand-in (tests may use the oracle; the
product's default is the HIP decode): what is covered here is sharding, the gather of
ragged shards and the file assignment, i.e. everything that differs for N > 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
import torbi_amd
from torbi_amd import distributed, synth


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def oracle_decode(observation, batch_frames, transition, initial):
    return torch.from_numpy(oracle.decode(observation.numpy(), batch_frames.numpy(),
                                          transition.numpy(), initial.numpy()))


def oracle_files(ins, outs, transition_file, initial_file, log_probs, gpu, num_threads, lengths=None):
    """Stand-in for core.from_files_to_files (one call per rank with that rank's files and their lengths)."""
    assert lengths is None or len(lengths) == len(ins)
    CALLS.append(len(ins))
    for k, (fin, fout) in enumerate(zip(ins, outs)):
        assert lengths is None or torch.load(fin).shape[0] == lengths[k]
        obs = torch.load(fin).unsqueeze(0)
        S = obs.shape[-1]
        idx = oracle_decode(obs, torch.tensor([obs.shape[1]], dtype=torch.int32),
                            torch.zeros(S, S), torch.zeros(S))
        torch.save(idx[0], fout)


CALLS = []


def worker(rank, size, port, tmp, B):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    r, s, _ = distributed.init_from_env(backend='gloo')
    assert (r, s) == (rank, size) and distributed.world() == (rank, size)
    T, S = 9, 12
    obs, trans, init = synth.problem(B, T, S, seed=2)
    frames = np.clip(synth.lengths(B, 1, T), 1, T)
    args = (torch.from_numpy(obs), torch.from_numpy(frames), torch.from_numpy(trans),
            torch.from_numpy(init))
    full = distributed.decode_sharded(*args, decode_fn=oracle_decode)
    want = oracle.decode(obs, frames, trans, init)
    assert np.array_equal(full.numpy(), want)
    local = distributed.decode_sharded(*args, gather=False, decode_fn=oracle_decode)
    lo, hi = distributed.shard_bounds(B, size, rank)
    assert np.array_equal(local.numpy(), want[lo:hi])
    # the rank holds only its own block
    mine = distributed.decode_sharded(args[0][lo:hi], args[1][lo:hi], args[2], args[3], decode_fn=oracle_decode, count=B)
    assert np.array_equal(mine.numpy(), want)

    ins = [os.path.join(tmp, f'in{k}.pt') for k in range(7)]
    outs = [os.path.join(tmp, f'out{k}.pt') for k in range(7)]
    lens = [3, 8, 2, 5, 9, 1, 4]
    if rank == 0:
        for f, n in zip(ins, lens):
            torch.save(torch.from_numpy(synth.scores(1, (n, 6), seed=n)), f)
    dist.barrier()
    saved = torbi_amd.BATCH_SIZE
    torbi_amd.core.BATCH_SIZE = 2
    n_mine = distributed.from_files_to_files(ins, outs, lengths=lens, decode_files=oracle_files)
    torbi_amd.core.BATCH_SIZE = saved
    assert CALLS == [n_mine]                      # ONE single-GPU call per rank, not one per batch
    counts = [None] * size
    dist.all_gather_object(counts, n_mine)
    assert sum(counts) == 7 and min(counts) >= 1
    for f, n in zip(outs, lens):
        assert torch.load(f).shape == (n,)
    dist.destroy_process_group()


@pytest.mark.parametrize('B', [5, 8])
def test_world_size_2_gloo(tmp_path, B):
    port = free_port()
    mp.spawn(worker, args=(2, port, str(tmp_path), B), nprocs=2, join=True)


def test_assign_files_keeps_whole_batches_and_puts_a_short_one_last():
    lengths = synth.lengths(1100, 100, 900).tolist()
    per_rank = distributed.assign_files(lengths, 512, 2)
    assert sorted(i for files in per_rank for i in files) == list(range(1100))
    for files in per_rank:
        for at in range(0, len(files), 512):
            block = files[at:at + 512]
            assert block == list(range(block[0], block[0] + len(block)))      # a whole loader batch, in order
    assert any(files[-1] == 1099 for files in per_rank)                       # the 76-file tail closes a rank's list


def test_single_process_paths():
    assert distributed.world() == (0, 1)
    x = torch.arange(6, dtype=torch.int32).reshape(3, 2)
    assert distributed.gather_indices(x, 3) is x
