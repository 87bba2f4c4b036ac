"""Peaked rows (bench.py's posteriorgram-like generator) on the cluster form, 2 and 4 batches per launch, by seeds per
item (TORBI_HIP_RESIDENT_KR).  GPU box."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S = 512, 200, 1440
trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
band = torch.from_numpy(synth.banded_transition(S, 87.2)).to(dev)
gen = torch.Generator(device=dev).manual_seed(7)
logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
flat = viterbi.fill_synthetic((B, T, S), 1, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
for name, data, matrix in (('peaked + dense', peaked, trans), ('peaked + band', peaked, band), ('flat + dense', flat, trans)):
    for n in (1, 2, 4):
        ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
        for _ in range(3):
            prof = []
            viterbi.decode_batches([data] * n, [frames] * n, matrix, init, workspaces=ws, path='cluster', _profile=prof)
        torch.cuda.synchronize()
        print(f'KR={os.environ.get("TORBI_HIP_RESIDENT_KR", "3")} {name:15s} x{n}: {1e3 * (prof[0] - prof[4]) / (T - 1):7.2f} us/step', flush=True)
