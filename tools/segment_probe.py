"""Two launch groups of n uniform batches (n = 10: 320 tiles each, more than the chip has CUs) on one stream and on two:
how well does the second group fill the first one's half-empty last round?  (GPU box; measured 43.8 ms on two streams
against 58.2 ms on one and 39 ms for a perfect 2.5 rounds.  Cutting the groups into two time segments -- tried, reverted --
gave 45.2 ms.)"""
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
