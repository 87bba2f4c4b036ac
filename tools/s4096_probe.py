"""Time-resident forms with 8-item tiles (2048 < S <= 4096) against the per-timestep pruned kernel. (GPU box)
    timeout 600 python tools/s4096_probe.py [S=4096] [T=200]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
for B, n in ((128, 1), (128, 2), (128, 4), (128, 8), (128, 16), (512, 1), (512, 4), (1024, 2)):
    obs = [viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=k, device=dev) for k in range(n)]
    frames = [torch.full((B,), T, dtype=torch.int32, device=dev)] * n
    ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
    ref, line = None, []
    for path in ('dense', 'cluster', 'resident'):
        for rep in range(2):
            prof = []
            got = viterbi.decode_batches(obs, frames, trans, init, workspaces=ws, path=path, _profile=prof)
        torch.cuda.synchronize()
        ref = got if ref is None else ref
        same = all(torch.equal(a, b) for a, b in zip(got, ref))
        fwd = (prof[0] - prof[4]) * (n if path == 'dense' else 1)        # the per-timestep path profiles its LAST batch only
        line.append(f'{path} [{viterbi.ROUTES[int(prof[3])]}] {1e3 * fwd / (T - 1):8.2f} us/step ({n * B * T / (fwd + prof[1] * (n if path == "dense" else 1)) / 1e3:6.2f} M/s) eq {same}')
    print(f'S={S} {n} x {B} items: ' + ' | '.join(line), flush=True)
