"""Child process of tests/test_gpu_scale.py::test_rccl_branch_with_one_rank (not collected by pytest).

Started FRESH (nothing in this process has touched a GPU before `init_from_env`), with WORLD_SIZE=1, RANK=0 and
TORBI_FORCE_DIST=1: torbi_amd.distributed then initialises the "nccl" backend (= RCCL on ROCm) for one rank and
every gather goes through `all_gather_into_tensor` on device tensors -- the multi-GPU code path of SURVEY 8(e),
exercised on the one GPU this box has.  Everything decoded is compared with the oracle.  Prints one JSON line.
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    import oracle
    import torbi_amd
    from torbi_amd import distributed, synth

    report = {}
    rank, size, local = distributed.init_from_env()
    assert dist.is_initialized(), 'TORBI_FORCE_DIST=1 must initialise torch.distributed with one rank'
    assert dist.get_backend() == 'nccl' and (rank, size) == (0, 1)
    dev = torch.device('cuda', torch.cuda.current_device())

    calls = {'all_gather': 0}
    plain_all_gather = dist.all_gather_into_tensor

    def counted(*args, **kwargs):
        calls['all_gather'] += 1
        return plain_all_gather(*args, **kwargs)

    dist.all_gather_into_tensor = counted

    # 1. decode_sharded(gather=True): this rank's block decoded, indices all-gathered on the device
    B, T, S = 96, 40, 360
    obs, trans, init = synth.problem(B, T, S, seed=5)
    frames = np.clip(synth.lengths(B, 1, T, seed=6), 1, T)
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    d_obs, d_frames, d_trans, d_init = (torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init))
    got = distributed.decode_sharded(d_obs, d_frames, d_trans, d_init, gather=True)
    torch.cuda.synchronize()
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), want), 'decode_sharded != oracle'
    assert calls['all_gather'] == 1, 'decode_sharded(gather=True) did not reach the collective'
    got = distributed.decode_sharded(d_obs, d_frames, d_trans, d_init, gather=True, count=B)   # shard handed over
    assert np.array_equal(got.cpu().numpy(), want) and calls['all_gather'] == 2
    report['decode_sharded'] = 'ok'

    # 2. the gather as a DecodePipeline `after=` hook: it runs on the pipeline's SIDE stream, behind the decode
    before = calls['all_gather']
    for group in (1, 3):
        pipe = torbi_amd.DecodePipeline(dev, depth=2, group=group)
        gathered, jobs = [], []

        def after(indices, sink=gathered):
            out = distributed.gather_indices(indices, indices.shape[0], force=True)
            sink.append(out)
            return out

        for k in range(5):
            b, t = 40 + 8 * k, 12 + 3 * k
            o = synth.scores(synth.STREAM_OBSERVATION, (b, t, S), seed=30 + k)
            f = np.clip(synth.lengths(b, 1, t, seed=k), 1, t)
            jobs.append((o, f, pipe.decode(torch.as_tensor(o).to(dev), torch.as_tensor(f).to(dev), d_trans, d_init,
                                           after=after)))
        pipe.synchronize()
        torch.cuda.synchronize()
        assert len(gathered) == 5
        for (o, f, idx), out in zip(jobs, gathered):
            w = oracle.decode(o, f, trans, init, num_threads=oracle.max_threads())
            assert np.array_equal(idx.cpu().numpy(), w) and np.array_equal(out.cpu().numpy(), w), f'group={group}'
    assert calls['all_gather'] == before + 10
    report['pipeline_after_hook'] = 'ok'

    # 3. distributed.from_files_to_files: 600 ragged files, this rank's batches in ONE single-GPU call, closing barrier
    count, S2 = 600, 256
    lengths = synth.lengths(count, 20, 180, seed=8).tolist()
    gen = torch.Generator().manual_seed(8)
    block = torch.rand(300, S2, generator=gen).mul_(6.0).log_softmax(-1)
    with tempfile.TemporaryDirectory(prefix='torbi_nccl_') as folder:
        ins, outs = [], []
        for k, n in enumerate(lengths):
            f = os.path.join(folder, f'in{k}.pt')
            start = (37 * k) % 100
            torch.save((block[start:start + n] + 0.01 * (k % 7)).log_softmax(-1).clone(), f)
            ins.append(f)
            outs.append(os.path.join(folder, f'out{k}.pt'))
        tf = os.path.join(folder, 'transition.pt')
        torch.save(torch.rand(S2, S2, generator=gen).mul_(4.0).softmax(-1), tf)
        done = distributed.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, lengths=lengths)
        assert done == count
        import math
        tiny = torch.finfo(torch.float32).tiny
        t_log = torch.log(torch.load(tf) + tiny).numpy()                      # reference core.py:341-347
        i_log = np.full((S2,), math.log(1. / S2 + tiny), dtype=np.float32)    # reference core.py:161-166
        for k in range(count):
            # the epsilon round trip on THIS device, then the oracle on the file alone
            x = torch.load(ins[k]).to(dev, dtype=torch.float32)
            x = torch.log(torch.exp(x) + tiny).cpu().numpy()[None]
            w = oracle.decode(x, [lengths[k]], t_log, i_log, num_threads=8)
            g = torch.load(outs[k])
            assert g.dtype == torch.int32 and tuple(g.shape) == (lengths[k],), f'file {k}'
            assert np.array_equal(g.numpy(), w[0]), f'file {k}'
    report['from_files_to_files'] = f'{count} files ok'
    dist.barrier()
    dist.destroy_process_group()
    report['all_gather_calls'] = calls['all_gather']
    print(json.dumps(report), flush=True)


if __name__ == '__main__':
    main()
