"""Per-phase timeline of pruned::step_pruned_kernel (instrumented build, -DPRUNED_STAMP).

    python tools/pruned_stamps.py build        # here (hipcc cross-compiles): tools/libtorbi_hip_stamp.so
    python tools/pruned_stamps.py [B T S]      # on the GPU box (default 512 40 1440; 128 40 4096 = configs[4] tiles)
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'libtorbi_hip_stamp.so')

if len(sys.argv) > 1 and sys.argv[1] == 'build':
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
                           '-fno-slp-vectorize', '-DPRUNED_STAMP', f'-I{ROOT}/include', '-o', LIB,
                           f'{ROOT}/torbi_amd/csrc/torbi_hip.hip'])
    sys.exit(0)

os.environ['TORBI_HIP_FORWARD'] = 'pruned'
import numpy as np, torch
import torbi_amd._lib as _lib
_lib.LIBRARY = LIB
import torbi_amd
B, T, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 40, 1440)))
dev = torch.device('cuda:0')
from torbi_amd import viterbi
obs = viterbi.fill_synthetic((B, T, S), 1, device=dev); trans = viterbi.fill_synthetic((S, S), 2, device=dev)
init = viterbi.fill_synthetic((S,), 3, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
viterbi.decode(obs, frames, trans, init); torch.cuda.synchronize()
lib = _lib.load()
KW, KS, NBLK = 12, 10, 256
buf = (ctypes.c_ulonglong * (NBLK * KW * KS))()
lib.torbi_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.torbi_hip_debug_stamps(buf, NBLK * KW * KS)
st = np.frombuffer(buf, dtype=np.uint64).reshape(NBLK, KW, KS).astype(np.float64)
d = np.diff(st, axis=2)
names = ['issue loads', 'merge + tile write', 'barrier', 'seed issue', 'block 0', 'fold seeds', 'scan', 'outputs+barrier', 'tile top lists']
span = (st[:, :, 9].max() - st[:, :, 0].min())
print('rc', rc, 'kernel span (ticks)', span)
for n, v in zip(names, d.mean(axis=(0, 1))):
    print(f'{n:20s} {v:9.0f} ticks  {100 * v / d.mean(axis=(0, 1)).sum():5.1f} %')
print('per-wave total', d.sum(axis=2).mean(), ' scan min/max over waves', d[:, :, 6].min(), d[:, :, 6].max())
