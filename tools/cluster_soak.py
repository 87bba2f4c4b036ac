"""Soak of the cluster form's self-validating exchange and of the segmented backtrace: the same batches decoded again and
again (cluster / band / small-state routes, ragged lengths), every result compared with the dense route's.  (GPU box)
    python tools/cluster_soak.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device('cuda:0')
cases = []
for (B, T, S, path) in [(512, 300, 1440, 'cluster'), (256, 200, 1440, 'cluster'), (40, 150, 1440, 'cluster'), (128, 120, 4096, 'cluster'),
                        (300, 100, 2048, 'cluster'), (512, 300, 1440, 'band'), (512, 200, 256, 'auto'), (2048, 100, 128, 'auto'),
                        (2048, 60, 1440, 'band'), (2048, 60, 1440, 'band-tiny'), (2100, 40, 1024, 'band'), (2560, 40, 1440, 'band-tiny'),
                        (512, 200, 1440, 'band-tiny')]:
    init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=B + S, device=dev)
    if path.startswith('band'):        # (2048+ items: whole tiles -- band_tile_forward.hpp; '-tiny': log(p + tiny), a constant outside the band)
        x = torch.arange(S, device=dev, dtype=torch.float32)
        tri = torch.clamp(87.2 - (x[:, None] - x[None, :]).abs(), min=0)
        p = tri / tri.sum(1, keepdim=True)
        trans = torch.log(p + torch.finfo(torch.float32).tiny) if path.endswith('tiny') else torch.log(p)
        path = 'auto' if path.endswith('tiny') else 'band'
    else:
        trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
    frames = torch.randint(1, T + 1, (B,), device=dev, dtype=torch.int32)
    frames[0] = T
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    ref = torbi_amd.decode(obs, frames, trans, init, path='dense')
    cases.append((B, T, S, path, obs, frames, trans, init, ws, ref))
t0 = time.time()
rounds = bad = gave_up = 0
side = torch.cuda.Stream(device=dev)
while time.time() - t0 < budget:
    for k, (B, T, S, path, obs, frames, trans, init, ws, ref) in enumerate(cases):
        stream = side if (rounds + k) & 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(stream):
            got = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path)
            same = torch.equal(got, ref)
            stats = viterbi.scan_stats(ws, B, T, S).cpu()
        bad += 0 if same else 1
        gave_up += int(stats[127])
    rounds += 1
print(f'{rounds} rounds x {len(cases)} cases in {time.time() - t0:.0f} s: {bad} mismatches, {gave_up} give-ups', flush=True)
sys.exit(1 if bad else 0)
