"""How the pruned pass responds to the shape of the posteriors (GPU box): the synthetic scores scaled by k spread the
observations over (-16k, 0]; log-softmax-like peaked rows prune earlier.  Prints forward ms and us per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi

B, T, S = 512, 200, 1440


def estimate_scan_depth(obs: torch.Tensor, trans: torch.Tensor, init: torch.Tensor, items: int = 8,
                         rows: int = 64, rank: int = 4) -> float:
    """How many entries of a sorted transition row the pruned pass would examine on this data: the
    95th percentile over a sample of (item, next-state) pairs, computed with torch ops on the exact
    posterior after one timestep (frames 0 and 1 of the first `items` items, `rows` evenly spaced
    next-states).  A cheap predictor that misses heavy tails (a few items with many dominant states) and steady
    state effects -- which is why the product reads the kernels' own scan statistics instead (viterbi._watch_resident)."""
    B, T, S = obs.shape
    items = max(1, min(items, B, (256 << 20) // (4 * S * S)))
    p = obs[:items, 0, :] + init[None, :]
    if T > 1:
        p = obs[:items, 1, :] + (p[:, None, :] + trans[None, :, :]).amax(dim=-1)
    pick = torch.linspace(0, S - 1, min(rows, S), device=trans.device).long()
    tj = trans[pick]                                                   # (rows, S)
    best = (p[:, None, :] + tj[None, :, :]).amax(dim=-1)               # (items, rows)
    thr = torch.topk(p, min(rank, S), dim=-1).values[:, -1]            # (items,)
    limit = best - thr[:, None]                                        # an entry t is examined while t > limit
    depth = (tj[None, :, :] > limit[:, :, None]).sum(dim=-1).float()   # (items, rows)
    depth = torch.nan_to_num(depth, nan=float(S))
    return float(torch.quantile(depth.flatten(), 0.95).item())



dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), 2, device=dev)
init = viterbi.fill_synthetic((S,), 3, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
base = viterbi.fill_synthetic((B, T, S), 1, device=dev)
cases = [('scores x1 (benchmark)', base), ('scores x4', base * 4), ('scores x8', base * 8), ('scores x12', base * 12), ('scores x16', base * 16), ('scores x64', base * 64),
         ('scores x1024', base * 1024), ('log_softmax(scores x4)', torch.log_softmax(base * 4, dim=-1)),
         ('scores x0.25', base * 0.25)]
cases += [('log_softmax(scores x16)', torch.log_softmax(base * 16, dim=-1)), ('one-hot-ish (x1 + 40 at a peak)', base + 40 * (base > -0.011))]
for name, obs in cases:
    print(f'{name:26s} estimated scan depth (95th pct of sampled pairs): {estimate_scan_depth(obs, trans, init):.0f} of {S}')
    for path in ('pruned', 'dense'):
        viterbi.set_forward_path(path)
        prof = []
        for _ in range(2):
            a = torbi_amd.decode(obs, frames, trans, init, workspace=ws, _profile=prof)
        if path == 'pruned':
            keep = a
        else:
            assert torch.equal(a, keep)
        print(f'{name:26s} {path:7s} forward {prof[0]:7.2f} ms  {prof[0] * 1e3 / prof[2]:6.2f} us/launch')
viterbi.set_forward_path('auto')

print('auto (measurement-based choice, 8 decodes each, last three timed):')
import time
for name, obs in cases:
    tr = trans.clone()                      # a fresh tensor: nothing known about it yet
    for i in range(8):
        if i == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        torbi_amd.decode(obs, frames, tr, init, workspace=ws)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'{name:26s} {dt * 1e3:6.2f} ms per decode; scan depth known to the host layer: {viterbi._known_depth(tr, S)}')
