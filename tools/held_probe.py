"""A handful of sequences: the held-matrix kernel (ONE launch, csrc/held_matrix_forward.hpp) against the per-timestep
kernels (generic trellis kernels = path 'dense' for B < 32; sorted-row scan = path 'pruned' for B <= 16).

    python tools/held_probe.py [S] [T]          # HELD_PROBE_ITEMS=1,2 HELD_PROBE_ONLY=1 TORBI_HIP_LIBRARY=... for A/B builds
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torbi_amd import synth, viterbi  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    dev = torch.device('cuda:0')
    print(f'S = {S}, T = {T}: ms per decode (forward ms, backtrace ms) [us per timestep]')
    for B in [int(x) for x in os.environ.get('HELD_PROBE_ITEMS', '1,2,3,4,6,8,12,16').split(',')]:
        obs, trans, init = synth.problem(B, T, S, seed=B)
        args = [torch.tensor(obs, device=dev), torch.full((B,), T, dtype=torch.int32, device=dev),
                torch.tensor(trans, device=dev), torch.tensor(init, device=dev)]
        space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
        row, ref = [f'B = {B:2d}'], None
        for name, path in (('held', 'held'),) if os.environ.get('HELD_PROBE_ONLY') else (('generic', 'dense'), ('rows', 'pruned'), ('held', 'held')):
            for _ in range(2):
                got = viterbi.decode(*args, workspace=space, path=path)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                viterbi.decode(*args, workspace=space, path=path)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            prof = []
            viterbi.decode(*args, workspace=space, path=path, _profile=prof)
            if ref is None:
                ref = got.cpu().numpy()
            same = np.array_equal(got.cpu().numpy(), ref)
            row.append(f'{name} {ms:6.3f} ({prof[0]:.3f} + {prof[1]:.3f}) [{ms * 1e3 / T:5.2f}]{"" if same else " DIFFERENT"}')
        stats = viterbi.scan_stats(space, B, T, S).cpu()
        print('   '.join(row))


if __name__ == '__main__':
    main()
