"""Per-phase cycle sums of resident::resident_forward_kernel (instrumented build, -DRESIDENT_STAMP).

    python tools/resident_stamps.py build        # here (hipcc cross-compiles): tools/libtorbi_hip_rstamp.so
    python tools/resident_stamps.py [batches] [frames]   # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get('STAMP_LIB') or os.path.join(ROOT, 'tools', 'libtorbi_hip_rstamp.so')

if len(sys.argv) > 1 and sys.argv[1] == 'build':
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
                           '-fno-slp-vectorize', '-DRESIDENT_STAMP', f'-I{ROOT}/include', '-o', LIB,
                           f'{ROOT}/torbi_amd/csrc/torbi_hip.hip'] + sys.argv[2:])
    sys.exit(0)

import numpy as np, torch
import torbi_amd._lib as _lib
_lib.LIBRARY = LIB
import torbi_amd
from torbi_amd import viterbi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B, S = 512, 1440
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), 2, device=dev)
init = viterbi.fill_synthetic((S,), 3, device=dev)
obs = [viterbi.fill_synthetic((B, T, S), 1, seed=k, device=dev) for k in range(n)]
frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
prof = []
PATH = os.environ.get('STAMP_PATH', 'resident')          # 'cluster': the cluster form (n = 1: R = 8 at 512 items)
for _ in range(2):
    viterbi.decode_batches(obs, frames, trans, init, path=PATH, _profile=prof)
torch.cuda.synchronize()
lib = _lib.load()
KW, KP = 12, 12
nwg = min(1024, n * B // 16 * (max(1, 256 // (n * B // 16)) if PATH == 'cluster' else 1))
buf = (ctypes.c_ulonglong * (nwg * 16 * KP))()
lib.torbi_hip_debug_phases.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.torbi_hip_debug_phases(buf, nwg * 16 * KP)
acc = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 16, KP)[:, :KW].astype(np.float64)
steps = T - 1
names = ['barrier A (wait for tile)', 'seed reads', 'pass head: loads -> block 0 + seeds', 'scan loop', 'outputs + top insert',
         'barrier B (wait for waves)', 'tile write + publish']
tot = acc[:, :, :7].sum(axis=2).mean()
print(f'rc {rc}; forward {prof[0]:.3f} ms for {n} batches x {T} frames; {tot / steps:.0f} ticks per timestep per wave '
      f'({prof[0] * 1e3 / steps:.1f} us)')
for i, name in enumerate(names):
    v = acc[:, :, i].mean() / steps
    print(f'{name:38s} {v:9.0f} ticks/step  {100 * v * steps / tot:5.1f} %')
if PATH == 'cluster':
    # (round 5's exchange: self-validating 16-byte pieces, no store drain, no flag -- the load that finds a piece is the hand-off)
    for i, name in zip((8, 10, 11), ('own row slices -> the other members (stores issued)',
                                     'the others\' slices + keys: asked for, awaited, -> tile / top lists', 'barrier E (tile complete)')):
        v = acc[:, :, i].mean() / steps
        print(f'{name:38s} {v:9.0f} ticks/step  (min / max over waves {acc[:, :, i].min() / steps:.0f} / {acc[:, :, i].max() / steps:.0f}; cluster, not in the percentages above)')
    print(f'polls that found a piece or key missing, per wave and timestep: {acc[:, :, 9].mean() / steps:.2f} '
          f'(0 = every first look succeeded)')
    per_wg = acc[:, :, 3].max(axis=1) / steps
    print(f'slowest wave\'s scan per workgroup: mean {per_wg.mean():.0f} ticks, max {per_wg.max():.0f}; '
          f'mean wave {acc[:, :, 3].mean() / steps:.0f} (ticks = shader cycles)')
if PATH != 'cluster':
    blocks = acc[:, :, 7].mean() / steps
    print(f'extra list blocks per wave and timestep: {blocks:.1f} over {np.ceil(90 / KW):.0f} passes -> '
          f'{16 * (1 + blocks / (90 / KW)):.0f} entries per row group on average')
print('scan ticks per wave: min/mean/max over waves', acc[:, :, 3].min() / steps, acc[:, :, 3].mean() / steps, acc[:, :, 3].max() / steps)
