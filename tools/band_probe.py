"""Band kernel (csrc/band_forward.hpp): parity against the oracle on small shapes, then timing on the bench's peaked rows +
pitch band at 512 x 500 x 1440 (one batch, a launch group of 8).  python tools/band_probe.py [--skip-parity]"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torbi_amd
from torbi_amd import synth, viterbi as v

dev = torch.device('cuda:0')


def banded(S, left, right, seed=0):
    _, trans, _ = synth.problem(1, 1, S, seed=seed)
    idx = np.arange(S)
    d = idx[None, :] - idx[:, None]            # i - j
    return np.where((d >= -left) & (d <= right), trans, -np.inf).astype(np.float32)


def parity():
    import oracle
    bad = 0
    cases = [(40, 12, 1440, 87, 87), (17, 9, 360, 10, 3), (64, 25, 360, 22, 22), (100, 7, 1440, 0, 0), (33, 11, 1024, 5, 60),
             (16, 6, 1440, 87, 87), (5, 8, 1440, 40, 40), (600, 5, 1440, 87, 87), (48, 9, 3072, 30, 30), (70, 6, 132, 8, 8),
             (260, 4, 1444, 86, 88), (24, 10, 512, 100, 100)]
    for (B, T, S, L, Rr) in cases:
        obs, _, init = synth.problem(B, T, S, seed=B + S)
        obs = np.round(obs * 2) / 2
        trans = np.round(banded(S, L, Rr, seed=S) * 2) / 2
        trans[S // 3] = -np.inf
        frames = np.clip(synth.lengths(B, 1, T, seed=3), 1, T).astype(np.int32)
        frames[0] = T
        want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
        args = [torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs, frames, trans, init)]
        prof = []
        got = torbi_amd.decode(*args, path='band', _profile=prof).cpu().numpy()
        ok = np.array_equal(got, want)
        bad += not ok
        print((B, T, S, L, Rr), 'route', v.ROUTES[int(prof[3])], 'OK' if ok else f'MISMATCH {np.sum(got != want)} of {got.size}',
              flush=True)
    return bad


def timing():
    B, T, S = 512, 500, 1440
    gen = torch.Generator(device=dev).manual_seed(7)
    logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
    centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
    logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
    peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
    del logits
    band = torch.from_numpy(synth.banded_transition(S, 87.2)).to(dev)
    init = torch.full((S,), math.log(1.0 / S), device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    ws = torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    outs = {}
    for path in ('band', 'dense', 'cluster'):
        for _ in range(2):
            out = torbi_amd.decode(peaked, frames, band, init, workspace=ws, path=path)
        torch.cuda.synchronize()
        prof = []
        torbi_amd.decode(peaked, frames, band, init, workspace=ws, path=path, _profile=prof)
        t0 = time.perf_counter()
        for _ in range(3):
            out = torbi_amd.decode(peaked, frames, band, init, workspace=ws, path=path)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / 3
        outs[path] = out.cpu().numpy()
        print(f'{path:8s} route {v.ROUTES[int(prof[3])]:8s} {sec * 1e3:7.3f} ms per decode = {B * T / sec / 1e6:6.1f} M timesteps/s; '
              f'forward {prof[0]:.3f} ms, backtrace {prof[1]:.3f} ms, preparation {prof[4]:.3f} ms', flush=True)
    print('band == dense:', np.array_equal(outs['band'], outs['dense']), ' band == cluster:', np.array_equal(outs['band'], outs['cluster']))
    stats = v.scan_stats(ws, B, T, S)
    spaces = [torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(8)]
    for path in ('band', 'auto'):
        for _ in range(2):
            got = v.decode_batches([peaked] * 8, [frames] * 8, band, init, workspaces=spaces, path=path)
        torch.cuda.synchronize()
        prof = []
        v.decode_batches([peaked] * 8, [frames] * 8, band, init, workspaces=spaces, path=path, _profile=prof)
        t0 = time.perf_counter()
        for _ in range(2):
            got = v.decode_batches([peaked] * 8, [frames] * 8, band, init, workspaces=spaces, path=path)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / 2
        print(f'group of 8, {path:5s} route {v.ROUTES[int(prof[3])]:8s} {sec * 1e3:7.3f} ms = {8 * B * T / sec / 1e6:6.1f} M timesteps/s; '
              f'forward {prof[0]:.3f} ms, backtrace {prof[1]:.3f} ms; first == single: '
              f'{np.array_equal(got[0].cpu().numpy(), outs["band"])}, last: {np.array_equal(got[7].cpu().numpy(), outs["band"])}', flush=True)


if __name__ == '__main__':
    bad = 0 if '--skip-parity' in sys.argv else parity()
    timing()
    print('mismatching cases:', bad)
