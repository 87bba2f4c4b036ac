"""Why does a workgroup's timestep take 70-75 us with 256 workgroups on the chip and 62.6 us with 32?  Same launch with
shared / distinct observation and history buffers (GPU box).  Measured: 75.3 us (8 distinct observation tensors and
histories), 69.5 us (one observation tensor read by all 8 batches), 67.5 us (one history as well): the streaming traffic
costs ~11 %, clocks / shared caches the rest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S, n = 512, 120, 1440, 8
trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
obs = [viterbi.fill_synthetic((B, T, S), 1, seed=k, device=dev) for k in range(n)]
frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
def run(o, w, label):
    prof = []
    for _ in range(3):
        viterbi.decode_batches(o, frames[:len(o)], trans, init, workspaces=w, path='resident', _profile=prof)
    print(f'{label}: forward {prof[0] - prof[4]:.3f} ms -> {(prof[0] - prof[4]) / (T - 1) * 1e3:.1f} us per workgroup step')
run(obs, ws, '8 distinct observation tensors, 8 histories')
run([obs[0]] * n, ws, 'one observation tensor 8 times, 8 histories')
run([obs[0]] * n, [ws[0]] * n, 'one observation tensor, ONE history written 8 times')
run(obs[:4], ws[:4], '4 batches (half the CUs)')
run(obs[:1], ws[:1], '1 batch (32 CUs)')
