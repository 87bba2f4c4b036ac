"""wide_forward_kernel against the oracle on awkward shapes (GPU box).
    TORBI_HIP_RESIDENT_KR=1 TORBI_HIP_WIDE=1 python tools/wide_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torbi_amd, oracle
from torbi_amd import synth, viterbi

dev = torch.device('cuda:0')
shapes = [(48, 6, 1444), (40, 5, 2048), (33, 9, 132), (64, 10, 360), (256, 4, 724), (270, 3, 1440), (48, 30, 1444), (17, 40, 64),
          (5, 12, 300), (100, 7, 1000)]
for (B, T, S) in shapes:
    for kind in ('plain', 'nearly_flat', 'full', 'minus_inf'):
        obs, trans, init = synth.problem(B, T, S, seed=41)
        if kind == 'nearly_flat':
            trans = (trans * np.float32(2 ** -12)).astype(np.float32)
        if kind == 'minus_inf':
            rng = np.random.default_rng(S)
            trans = np.where(rng.random((S, S)) < 0.7, np.float32(-np.inf), trans).astype(np.float32)
            obs = np.where(rng.random(obs.shape) < 0.2, np.float32(-np.inf), obs).astype(np.float32)
        frames = synth.lengths(B, 1, T, seed=S) if kind != 'full' else np.full(B, T, np.int32)
        want = oracle.decode(obs, frames, trans, init, mode=1)
        args = [torch.as_tensor(x).to(dev) for x in (obs, np.asarray(frames, np.int32), trans, init)]
        got = torbi_amd.decode(*args, path='resident').cpu().numpy()
        bad = np.argwhere(got != want)
        print((B, T, S), kind, viterbi.last_forward_kernel(), 'ok' if len(bad) == 0 else f'{len(bad)} differ; first {bad[:4].tolist()}', flush=True)
