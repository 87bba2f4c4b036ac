"""Peaked rows and the benchmark on the time-resident forms and the per-timestep paths: launch groups and one batch.
(GPU box.)  Written for the threshold-seed experiment of round 3 (the TORBI_HIP_WIDE_SEEDS build it compared
is not in the tree any more; profiles/r03_threshold_seeds_probe.txt holds its output); still useful as a data-dependence probe."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S, n = 512, 200, 1440, 8
trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
gen = torch.Generator(device=dev).manual_seed(7)
frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
def run(name, data, paths=('resident', 'cluster1', 'pruned1', 'dense1')):
    ref = None
    for path in paths:
        single = path.endswith('1')
        p = path.rstrip('1')
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if single:
                got = [torbi_amd.decode(data, frames[0], trans, init, workspace=ws[0], path=p)]
            else:
                got = viterbi.decode_batches([data] * n, frames, trans, init, workspaces=ws, path=p)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ref = got[0] if ref is None else ref
        count = 1 if single else n
        stats = viterbi.scan_stats(ws[0], B, T, S, path='resident') if p in ('resident', 'cluster') else None
        depth = f', {viterbi.critical_blocks(stats):.1f} list blocks per pass' if stats is not None else ''
        print(f'{name}: {path:9s} {count * B * T / dt / 1e6:7.2f} M timesteps/s ({dt * 1e3:.1f} ms){depth}, same indices {torch.equal(got[0], ref)}', flush=True)
for width in (12.0, 3.0, 40.0):
    logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
    centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
    logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / width) ** 2
    peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
    del logits
    run(f'peak width {width}', peaked)
base = viterbi.fill_synthetic((B, T, S), 1, device=dev)
run('benchmark', base, paths=('resident', 'cluster1', 'pruned1'))
run('benchmark x16', base * 16.0, paths=('resident', 'cluster1', 'dense1'))
