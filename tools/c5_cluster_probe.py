"""configs[4] (128 x T x 4096) on the cluster form, per variant library.  python tools/c5_cluster_probe.py [T=300]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, S = 128, 4096
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=5, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
ref = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path='pruned')
for path in ('pruned', 'cluster'):
    for few in (False, True):
        if path == 'pruned' and few:
            continue
        viterbi._depth_record(trans, S)[0] = 1.0 if few else None
        for rep in range(3):
            prof = []
            got = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path, _profile=prof)
        torch.cuda.synchronize()
        print(f'c5 {path:8s} few_seeds={few}: {1e3 * (prof[0] - prof[4]) / (T - 1):7.2f} us/step, backtrace {prof[1]:.2f} ms, '
              f'{B * T / (prof[0] + prof[1]) / 1e3:.2f} M timesteps/s, equal {torch.equal(got, ref)}', flush=True)
