import os, sys, math
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S = 512, 500, 1440
gen = torch.Generator(device=dev).manual_seed(7)
logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
for hw in (12.0, 50.0, 100.0):
    band = torch.from_numpy(synth.banded_transition(S, hw)).to(dev)
    for path in ('dense', 'cluster'):
        for _ in range(3):
            prof = []
            viterbi.decode(peaked, frames, band, init, workspace=ws, path=path, _profile=prof)
        print(f'half width {hw:5.0f} {path:8s}: forward {prof[0]:6.2f} ms (prep {prof[4]:.2f}), backtrace {prof[1]:5.2f} ms')
