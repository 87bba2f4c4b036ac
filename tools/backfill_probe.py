"""How does the GPU place the workgroups of a time-resident launch when some finish early?  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S = 512, 200, 1440
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)

def group(n, seed):
    obs = [viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=seed + k, device=dev) for k in range(n)]
    frames = [torch.tensor(synth.lengths(B, T // 9, T, seed=seed + k), device=dev) for k in range(n)]
    ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
    return obs, frames, ws

def run(jobs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (obs, frames, ws), stream, asc in jobs:
        with torch.cuda.stream(stream):
            viterbi.decode_batches(obs, frames, trans, init, workspaces=ws, path='resident', shortest_first=asc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
a, b, c, d = group(8, 0), group(8, 100), group(8, 200), group(8, 300)
big = (a[0] + b[0], a[1] + b[1], a[2] + b[2])
for rep in range(2):
    print('one ragged group of 8 (256 workgroups), longest first %.2f ms, shortest first %.2f ms' % (run([(a, s1, False)]), run([(a, s1, True)])))
    print('one ragged group of 16 (512 workgroups), longest first %.2f ms, shortest first %.2f ms' % (run([(big, s1, False)]), run([(big, s1, True)])))
    print('two groups of 8 on two streams: long/long %.2f  short/short %.2f  short/long %.2f  long/short %.2f ms' % (
        run([(a, s1, False), (b, s2, False)]), run([(a, s1, True), (b, s2, True)]),
        run([(a, s1, True), (b, s2, False)]), run([(a, s1, False), (b, s2, True)])))
    print('four groups of 8 on two streams: all long %.2f  alternating short/long %.2f ms' % (
        run([(a, s1, False), (b, s2, False), (c, s1, False), (d, s2, False)]),
        run([(a, s1, True), (b, s2, False), (c, s1, True), (d, s2, False)])))
