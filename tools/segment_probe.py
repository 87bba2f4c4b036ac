"""Two launch groups of 10 uniform batches (320 tiles each, two time segments) on two streams: do they interleave? (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S = 512, 200, 1440
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
def group(seed):
    obs = [viterbi.fill_synthetic((B, T, S), 1, seed=seed + k, device=dev) for k in range(n)]
    frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
    ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
    return obs, frames, ws
def run(jobs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for (obs, frames, ws), stream in jobs:
        with torch.cuda.stream(stream):
            viterbi.decode_batches(obs, frames, trans, init, workspaces=ws, path='resident')
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
a, b = group(0), group(100)
run([(a, s1), (b, s2)])
for _ in range(2):
    print(f'{n} batches per group: one group {run([(a, s1)]):.2f} ms; two on one stream {run([(a, s1), (b, s1)]):.2f} ms; '
          f'two on two streams {run([(a, s1), (b, s2)]):.2f} ms')
