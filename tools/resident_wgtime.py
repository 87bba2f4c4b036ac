"""When do the workgroups of a ragged time-resident launch start and end?  (instrumented build: python
tools/resident_stamps.py build; GPU box: python tools/resident_wgtime.py [batches])"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torbi_amd._lib as _lib
_lib.LIBRARY = os.path.join(ROOT, 'tools', 'libtorbi_hip_rstamp.so')
from torbi_amd import viterbi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, T, S = 512, 200, 1440
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
obs = [viterbi.fill_synthetic((B, T, S), 1, seed=k, device=dev) for k in range(n)]
frames = [torch.tensor(synth.lengths(B, T // 9, T, seed=k), device=dev) for k in range(n)]
for _ in range(2):
    viterbi.decode_batches(obs, frames, trans, init, path='resident')
torch.cuda.synchronize()
lib = _lib.load()
nwg = n * B // 16
buf = (ctypes.c_ulonglong * (4 * nwg))()
lib.torbi_hip_debug_wgtime.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.torbi_hip_debug_wgtime(buf, 4 * nwg)
a = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 4).astype(np.float64)
t0 = a[:, 0].min()
start, end, steps = (a[:, 0] - t0) / 1e5, (a[:, 1] - t0) / 1e5, a[:, 2]
print(f'{nwg} workgroups; launch spans {end.max():.2f} ms; per step {np.median((end - start) / np.maximum(steps - 1, 1)) * 1e3:.1f} us')
for lo in range(0, nwg, 32):
    sl = slice(lo, lo + 32)
    print(f'wgs {lo:4d}-{lo + 31:4d}: steps {int(steps[sl].max()):4d}..{int(steps[sl].min()):4d}  start {start[sl].min():7.2f}..{start[sl].max():7.2f} ms  '
          f'end {end[sl].min():7.2f}..{end[sl].max():7.2f} ms')
