"""A variant build of the time-resident kernel against the oracle on awkward shapes (GPU box).
    TORBI_HIP_LIBRARY=tools/libtorbi_hip_X.so TORBI_HIP_RESIDENT_KR=1 python tools/cap_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torbi_amd, oracle
from torbi_amd import synth, viterbi

dev = torch.device('cuda:0')
shapes = [(48, 6, 1444), (40, 5, 2048), (33, 9, 132), (64, 10, 360), (256, 4, 724), (270, 3, 1440), (48, 30, 1444), (17, 40, 64),
          (5, 12, 300), (100, 7, 1000)]
bad_total = 0
for (B, T, S) in shapes:
    for kind in ('plain', 'nearly_flat', 'flat', 'full', 'minus_inf', 'peaked'):
        obs, trans, init = synth.problem(B, T, S, seed=41)
        rng = np.random.default_rng(S)
        if kind == 'nearly_flat':
            trans = (trans * np.float32(2 ** -12)).astype(np.float32)
        if kind == 'flat':
            trans = np.full((S, S), np.float32(-1.25))
        if kind == 'minus_inf':
            trans = np.where(rng.random((S, S)) < 0.7, np.float32(-np.inf), trans).astype(np.float32)
            obs = np.where(rng.random(obs.shape) < 0.2, np.float32(-np.inf), obs).astype(np.float32)
        if kind == 'peaked':
            centre = rng.integers(0, S, size=(B, T, 1))
            obs = (obs - ((np.abs(np.arange(S)[None, None, :] - centre) / 6.0) ** 2)).astype(np.float32)
        frames = synth.lengths(B, 1, T, seed=S) if kind != 'full' else np.full(B, T, np.int32)
        want = oracle.decode(obs, frames, trans, init, mode=1)
        args = [torch.as_tensor(x).to(dev) for x in (obs, np.asarray(frames, np.int32), trans, init)]
        got = torbi_amd.decode(*args, path='resident').cpu().numpy()
        bad = np.argwhere(got != want)
        bad_total += len(bad)
        if len(bad):
            print((B, T, S), kind, viterbi.last_forward_kernel(), f'{len(bad)} differ; first {bad[:4].tolist()}', flush=True)
print('checked', len(shapes) * 6, 'cases; mismatching entries:', bad_total)
