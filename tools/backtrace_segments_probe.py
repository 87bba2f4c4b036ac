"""Backtrace of ONE batch behind a time-resident forward launch: whole paths (TORBI_HIP_BACKTRACE_SEGMENTS=1) against K
speculative segments per path (lazy_backtrace.hpp, chase_segment / stitch_segments), child process per setting.
    python tools/backtrace_segments_probe.py [K ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
import torbi_amd
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
import math
for (B, T, S, path) in [(512, 500, 1440, 'cluster'), (64, 500, 1440, 'cluster'), (512, 500, 1440, 'band'), (512, 500, 1440, 'band-peaked'),
                        (512, 500, 1440, 'band-smooth'), (512, 500, 1440, 'band-mixed'), (128, 300, 4096, 'cluster')]:
    init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
    kind, path = path, path.split('-')[0]
    if kind != path:      # posteriorgram-like rows: bench.py's (a random centre per frame) / a centre that wanders (a pitch track)
        gen = torch.Generator(device=dev).manual_seed(7)
        logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
        if kind.endswith('peaked'):
            centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
        else:
            centre = (S / 2 + torch.cumsum(torch.randn((B, T, 1), device=dev, generator=gen) * 12.0, dim=1)).remainder(S).long()
        logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
        if kind.endswith('mixed'):      # stretches of 50 frames without a peak (unvoiced), 50 with one
            flat = ((torch.arange(T, device=dev) // 50) %% 2 == 1)[None, :, None]
            logits = torch.where(flat, torch.randn((B, T, S), device=dev, generator=gen) * 0.3, logits)
        obs = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
    if path == 'band':
        x = torch.arange(S, device=dev, dtype=torch.float32)
        tri = torch.clamp(87.2 - (x[:, None] - x[None, :]).abs(), min=0)
        trans = torch.log(tri / tri.sum(1, keepdim=True))
    else:
        trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    frames[::3] = T - 37
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    ref = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path='dense')
    best = None
    for rep in range(5):
        prof = []
        got = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path, _profile=prof)
        best = prof[1] if best is None else min(best, prof[1])
    print(f'{B} x {T} x {S} {kind} [{viterbi.ROUTES[int(prof[3])]}]: backtrace {best:.3f} ms  equal {torch.equal(got, ref)}', flush=True)
'''
for k in sys.argv[1:] or ['1', '4', '8', '16']:
    print(f'== TORBI_HIP_BACKTRACE_SEGMENTS={k}', flush=True)
    subprocess.run([sys.executable, '-c', CHILD % ROOT], env=dict(os.environ, TORBI_HIP_BACKTRACE_SEGMENTS=k))
