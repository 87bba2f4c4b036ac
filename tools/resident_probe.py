"""Time-resident forward kernel vs per-timestep launches on the headline shape (GPU box).
    python tools/resident_probe.py [batches] [frames] [B] [S]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 120
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1440
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
obs = [viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=k, device=dev) for k in range(n)]
frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]

ref = [torbi_amd.decode(obs[k], frames[k], trans, init, workspace=ws[k], path='cluster') for k in range(n)]
torch.cuda.synchronize()
prof = []
torbi_amd.decode(obs[0], frames[0], trans, init, workspace=ws[0], path='cluster', _profile=prof)
print(f'cluster form, one batch: forward {prof[0]:.3f} ms ({1e3 * prof[0] / max(prof[2], 1):.2f} us/launch), '
      f'backtrace {prof[1]:.3f} ms')

for m in sorted({1, 2, 4, n}):
    for rep in range(2):
        prof = []
        got = viterbi.decode_batches(obs[:m], frames[:m], trans, init, workspaces=ws[:m], path='resident', _profile=prof)
    ok = all(torch.equal(a, b) for a, b in zip(got, ref[:m]))
    steps = m * B * T
    print(f'resident x{m}: forward {prof[0]:.3f} ms (prep {prof[4]:.3f}), backtrace {prof[1]:.3f} ms, '
          f'{steps / (prof[0] + prof[1]) / 1e3:.2f} M timesteps/s, per 512-item step {1e3 * (prof[0] - prof[4]) / (T - 1) / m * (512 / B):.2f} us, '
          f'equal to per-step path: {ok}')
