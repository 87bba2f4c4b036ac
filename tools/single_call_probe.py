"""One decode call of one batch in the cluster form (GPU box): forward us per timestep, backtrace, equality with the
per-timestep path.  Environment: TORBI_HIP_LIBRARY (variant build), TORBI_HIP_CLUSTER_R, TORBI_HIP_CLUSTER_KW6.
    python tools/single_call_probe.py [T] [S] [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1440
Bs = [int(x) for x in sys.argv[3:]] or [512]
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
for B in Bs:
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    ref = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path='dense')
    best = None
    for rep in range(4):
        prof = []
        got = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path='cluster', _profile=prof)
        us = 1e3 * (prof[0] - prof[4]) / (T - 1)
        best = us if best is None else min(best, us)
    stats = viterbi.scan_stats(ws, B, T, S, path='resident').cpu()
    print(f'B={B} S={S} T={T}: cluster [{viterbi.ROUTES[int(prof[3])]}] {viterbi.last_forward_kernel()} {best:7.2f} us/step '
          f'bt {prof[1]:.3f} ms  equal {torch.equal(got, ref)}  gave_up {int(stats[127])}', flush=True)
