"""Forward us per timestep of the cluster form for a few launch shapes (A/B of builds: TORBI_HIP_LIBRARY=...).
    python tools/cluster_poll_probe.py [S] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
out = []
for B, n in ((64, 1), (256, 1), (512, 1), (768, 1), (512, 2), (512, 4)):
    obs = [viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=k, device=dev) for k in range(n)]
    frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
    ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
    best = 1e9
    for rep in range(4):
        prof = []
        viterbi.decode_batches(obs, frames, trans, init, workspaces=ws, path='cluster', _profile=prof)
        torch.cuda.synchronize()
        best = min(best, 1e3 * (prof[0] - prof[4]) / (T - 1))
    out.append(f'{n}x{B}: {best:6.2f}')
print('  '.join(out), flush=True)
