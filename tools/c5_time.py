"""BASELINE configs[4] (128 x 2000 x 4096) decoded a few times: the command behind profiles/r06_c5_* (tools/collect_profiles.sh:
PROFILE_CMD='python3 tools/c5_time.py').  python tools/c5_time.py [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torbi_amd
from torbi_amd import synth, viterbi as v
dev = torch.device('cuda:0')
B, T, S = 128, 2000, 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
obs = v.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=5, device=dev)
trans = v.fill_synthetic((S, S), synth.STREAM_TRANSITION, seed=0, device=dev)
init = v.fill_synthetic((S,), synth.STREAM_INITIAL, seed=0, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
ws = torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
best = None
for _ in range(n):
    prof = []
    torbi_amd.decode(obs, frames, trans, init, workspace=ws, _profile=prof)
    best = prof if best is None or prof[0] < best[0] else best
print(f'{v.last_forward_kernel()} route {v.ROUTES[int(best[3])]}: forward {best[0]:.2f} ms ({(best[0] - best[4]) * 1e3 / (T - 1):.2f} us per timestep), '
      f'backtrace {best[1]:.2f} ms, preparation {best[4]:.2f} ms; {B * T / (best[0] + best[1]) / 1e3:.2f} M timesteps/s')
