"""The reference's evaluation call as it really is (torbi/evaluate/core.py:97-103 -> core.py:341-347): the pitch transition goes
through log(p + tiny), so it is log(tiny) = -87.34 outside the band, NOT -inf.  What AUTO does with it today, a launch group of
N x 512 x T x 1440 peaked rows, beside the -inf band.  python tools/pitch_tiny_probe.py [N] [T]"""
import math, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torbi_amd
from torbi_amd import synth, viterbi as v
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
B, S = 512, 1440
gen = torch.Generator(device=dev).manual_seed(7)
kinds = {}
logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
kinds['peaked (random centre per frame)'] = logits - ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
centre = (S / 2 + torch.cumsum(torch.randn((B, T, 1), device=dev, generator=gen) * 12.0, dim=1)).remainder(S).long()
kinds['smooth (a wandering centre), tails like a network softmax'] = (logits - ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2).clamp_(min=-30.0)
del logits
tiny = torch.finfo(torch.float32).tiny
x = np.arange(S)
tri = np.clip(87.2 - np.abs(x[:, None] - x[None, :]), 0, None).astype(np.float32)
probs = torch.from_numpy(tri / tri.sum(axis=1, keepdims=True)).to(dev)
mats = {'log(p + tiny)  [evaluate: log_probs=True]': torch.log(probs + tiny), 'log(p)  [-inf outside the band]': torch.log(probs)}
init = torch.full((S,), math.log(1.0 / S), device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
spaces = [torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(N)]
for kname, lg in kinds.items():
    obs = torch.log_softmax(lg, dim=-1).clamp_(min=math.log(tiny))
    for mname, band in mats.items():
        best = None
        for _ in range(4):
            prof = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            got = v.decode_batches([obs] * N, [frames] * N, band, init, workspaces=spaces, _profile=prof)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = (dt, prof) if best is None or dt < best[0] else best
        dt, prof = best
        print(f'{kname} | {mname}: route {v.ROUTES[int(prof[3])]} {v.last_forward_kernel()} {dt * 1e3:.2f} ms = {N * B * T / dt / 1e6:.1f} M timesteps/s '
              f'(forward {prof[0]:.2f}, backtrace {prof[1]:.2f})', flush=True)
