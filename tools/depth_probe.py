"""Scan depth (list blocks per wave pass on the critical path, viterbi.critical_blocks) of time-resident launches by
form and seeds per item -- the quantity FEW_SEEDS_GATE / RESIDENT_GATE in torbi_amd/viterbi.py are compared with.

    python tools/depth_probe.py            # benchmark rows and peaked rows, 512 x 500 x 1440, per-timestep kernels beside
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torbi_amd import synth, viterbi  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    B, T, S = 512, 500, 1440
    obs, trans, init = synth.problem(B, T, S, seed=0)
    rng = np.random.default_rng(1)
    centre = rng.integers(0, S, size=(B, T, 1))
    peaked = (obs - ((np.abs(np.arange(S)[None, None, :] - centre) / 6.0) ** 2)).astype(np.float32)
    band = synth.banded_transition(S, 12.0)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    d_init = torch.tensor(init, device=dev)
    space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    print(f'{"rows":>10} {"matrix":>8} {"path":>9} {"seeds":>6} {"blocks":>8} {"of S/16":>8} {"us/step":>8}')
    for rows, data in (('benchmark', obs), ('peaked', peaked)):
        d_obs = torch.tensor(data, device=dev)
        for mname, matrix in (('dense', trans), ('band', band)):
            d_trans = torch.tensor(matrix, device=dev)
            for path in ('dense', 'pruned', 'resident', 'cluster'):
                for flag, seeds in ((512, 1), (1024, 3)) if path in viterbi.TIME_RESIDENT else ((0, 0),):
                    viterbi._depth_record(d_trans, S)[0] = 1.0 if flag == 512 else float(S)
                    for _ in range(2):
                        viterbi.decode(d_obs, frames, d_trans, d_init, workspace=space, path=path)
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    viterbi.decode(d_obs, frames, d_trans, d_init, workspace=space, path=path)
                    b.record()
                    torch.cuda.synchronize()
                    blocks = viterbi.critical_blocks(viterbi.scan_stats(space, B, T, S).cpu()) if seeds else 0.0
                    print(f'{rows:>10} {mname:>8} {path:>9} {seeds:>6} {blocks:8.2f} {blocks / (S / 16):8.3f} '
                          f'{a.elapsed_time(b) * 1e3 / T:8.2f}')


if __name__ == '__main__':
    main()
