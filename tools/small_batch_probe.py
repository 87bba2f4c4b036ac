"""Small batches (B <= 16): the sorted-row step kernel against the generic row kernels.  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
T = 500
for B, S in ((1, 1440), (4, 1440), (16, 1440), (1, 360), (1, 4096), (8, 4096)):
    obs = viterbi.fill_synthetic((B, T, S), 1, device=dev)
    trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    out = {}
    for path in ('auto', 'dense'):            # B <= 16: auto -> rows, dense -> generic row kernels
        for _ in range(2):
            res = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            res = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        prof = []
        torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path, _profile=prof)
        out[path] = (dt, res, viterbi.ROUTES[int(prof[3])], prof[0], prof[1])
    a, g = out['auto'], out['dense']
    print(f'B={B:2d} S={S}: {a[2]} {a[0] * 1e3:.3f} ms ({a[0] / (T - 1) * 1e6:.2f} us/step; forward {a[3]:.3f} + backtrace {a[4]:.3f} ms)   '
          f'{g[2]} {g[0] * 1e3:.3f} ms ({g[0] / (T - 1) * 1e6:.2f} us/step; forward {g[3]:.3f} + backtrace {g[4]:.3f} ms)   equal: {torch.equal(a[1], g[1])}')
