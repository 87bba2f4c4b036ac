"""Cluster form against the per-timestep paths over batch sizes (GPU box): us per timestep, forward only.
    timeout 300 python tools/cluster_sweep.py [S] [T]"""
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
for B in (1, 4, 8, 16, 17, 32, 64, 128, 192, 256, 384, 512, 768, 1024):
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    line, ref = [], None
    for path in ('auto', 'dense', 'cluster'):
        for rep in range(3):
            prof = []
            got = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path, _profile=prof)
        torch.cuda.synchronize()
        ref = got if ref is None else ref
        line.append(f'{path} [{viterbi.ROUTES[int(prof[3])]}] {1e3 * (prof[0] - prof[4]) / (T - 1):7.2f} us/step bt {prof[1]:.3f} ms eq {torch.equal(got, ref)}')
    print(f'B={B:5d} S={S}: ' + ' | '.join(line), flush=True)
