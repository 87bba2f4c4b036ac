"""Forward / backtrace split of one decode: python tools/phase_probe.py B T S [path]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth
B, T, S = (int(x) for x in sys.argv[1:4])
path = sys.argv[4] if len(sys.argv) > 4 else None
dev = torch.device('cuda:0')
obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
for _ in range(3):
    prof = []
    torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path, _profile=prof)
print(f'{B} x {T} x {S} ({viterbi.ROUTES[int(prof[3])]}): forward {prof[0]:.3f} ms ({1e3 * prof[0] / max(prof[2], 1):.2f} us per launch, '
      f'{int(prof[2])} launches), argmax + backtrace {prof[1]:.3f} ms ({1e3 * prof[1] / max(T - 1, 1):.2f} us per step)')
