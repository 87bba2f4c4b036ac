"""One 512 x T x 1440 batch on the per-timestep, cluster and whole-tile forms (per variant library). python tools/serial_probe.py [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, S = 512, 1440
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
ref = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path='pruned')
for path in ('pruned', 'cluster'):
    for rep in range(3):
        prof = []
        got = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path, _profile=prof)
    torch.cuda.synchronize()
    print(f'{path:8s}: {1e3 * (prof[0] - prof[4]) / (T - 1):7.2f} us/step, backtrace {prof[1]:.3f} ms, equal {torch.equal(got, ref)}', flush=True)
