"""Launch groups on data where the pruning bound does not bite (peaked rows + dense random transition). (GPU box)"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S, n = 512, 200, 1440, 8
trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
gen = torch.Generator(device=dev).manual_seed(7)
frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
for width in (12.0, 3.0, 40.0):
    logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
    centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
    logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / width) ** 2
    peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
    del logits
    for path in ('resident', 'dense', 'pruned'):
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            viterbi.decode_batches([peaked] * n, frames, trans, init, workspaces=ws, path=path)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'peak width {width}: {path:9s} {n * B * T / dt / 1e6:7.2f} M timesteps/s  ({dt * 1e3:.1f} ms per group)')
# observation spread sweep on the synthetic benchmark inputs (DESIGN.md 4.5 for the per-timestep kernels)
base = viterbi.fill_synthetic((B, T, S), 1, device=dev)
for scale in (1.0, 4.0, 16.0, 64.0):
    scaled = base * scale
    for path in ('resident', 'dense'):
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            viterbi.decode_batches([scaled] * n, frames, trans, init, workspaces=ws, path=path)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        stats = viterbi.scan_stats(ws[0], B, T, S, path='resident') if path == 'resident' else None
        depth = f', {viterbi.critical_blocks(stats):.1f} of {S // 16} list blocks per scan' if stats is not None else ''
        print(f'observations x{scale}: {path:9s} {n * B * T / dt / 1e6:7.2f} M timesteps/s{depth}')
