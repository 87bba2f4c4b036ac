"""Do two time-resident launch groups on two HIP streams share the chip?  (GPU box)
    python tools/two_stream_groups.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torbi_amd
from torbi_amd import viterbi, synth

dev = torch.device('cuda:0')
B, T, S = 512, 200, 1440
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)


def group(n, ragged, seed):
    obs = [viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=seed + k, device=dev) for k in range(n)]
    if ragged:
        frames = [torch.tensor(synth.lengths(B, T // 9, T, seed=seed + k), device=dev) for k in range(n)]
    else:
        frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
    ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
    return obs, frames, ws


def run(groups, streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (obs, frames, ws), stream in zip(groups, streams):
        with torch.cuda.stream(stream):
            viterbi.decode_batches(obs, frames, trans, init, workspaces=ws, path='resident')
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for n, ragged in ((4, False), (8, False), (8, True)):
    a, b = group(n, ragged, 0), group(n, ragged, 100)
    run([a, b], [s1, s2])
    one = run([a], [s1])
    same = run([a, b], [s1, s1])
    two = run([a, b], [s1, s2])
    print(f'{n} batches per group, ragged={ragged}: one group {one:.2f} ms; two groups on one stream {same:.2f} ms; '
          f'on two streams {two:.2f} ms')
