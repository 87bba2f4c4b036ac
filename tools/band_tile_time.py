"""Forward time of the whole-tile band kernel (csrc/band_tile_forward.hpp): a launch group of N x 512 x T x 1440, pitch band,
peaked rows; the delivered clock from the kernel's own statistics.  python tools/band_tile_time.py [N] [T]
(tools/variants_probe.py script: PROBE_SCRIPT=band_tile_time.py)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torbi_amd
from torbi_amd import synth, viterbi as v
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
B, S = 512, 1440
gen = torch.Generator(device=dev).manual_seed(7)
logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
del logits
band = torch.from_numpy(synth.banded_transition(S, 87.2, tiny=bool(os.environ.get('BAND_TILE_TINY')))).to(dev)      # (BAND_TILE_TINY=1: log(p + tiny))
init = torch.full((S,), math.log(1.0 / S), device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
spaces = [torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(N)]
best = None
for _ in range(5):
    prof = []
    v.decode_batches([peaked] * N, [frames] * N, band, init, workspaces=spaces, path='auto', _profile=prof)
    best = prof if best is None or prof[0] < best[0] else best
stats = v.scan_stats(spaces[0], B, T, S)
ghz = float(stats[120]) / max(float(stats[121]), 1.0) * 0.1
us = best[0] * 1e3 / (T - 1)
cells = N * B * S * 176.0 / (us * 1e-6)
print(f'{v.last_forward_kernel()} route {v.ROUTES[int(best[3])]} forward {best[0]:.3f} ms = {us:.2f} us per timestep at {ghz:.2f} GHz; '
      f'{N * B * (T - 1) / best[0] / 1e3:.1f} M timesteps/s forward; {cells / 1e12:.1f} Tcell/s = {100 * cells / (16384 * ghz * 1e9):.1f} % of the ALU ceiling; '
      f'backtrace {best[1]:.3f} ms; give-ups {int(stats[127])}')
