"""Forward time of the band kernel alone: 512 x 500 x 1440, pitch band, peaked rows (tools/variants_probe.py script)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torbi_amd
from torbi_amd import synth, viterbi as v
dev = torch.device('cuda:0')
B, T, S = 512, 500, 1440
gen = torch.Generator(device=dev).manual_seed(7)
logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
band = torch.from_numpy(synth.banded_transition(S, 87.2)).to(dev)
init = torch.full((S,), math.log(1.0 / S), device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
ws = torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
best = None
for _ in range(6):
    prof = []
    torbi_amd.decode(peaked, frames, band, init, workspace=ws, path='band', _profile=prof)
    best = prof if best is None or prof[0] < best[0] else best
print(f'route {v.ROUTES[int(best[3])]} forward {best[0]:.3f} ms = {best[0] * 1e3 / (T - 1):.2f} us per timestep; backtrace {best[1]:.3f} ms')
