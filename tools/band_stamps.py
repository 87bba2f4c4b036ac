"""Per-phase cycle sums of band::band_forward_kernel (instrumented build, -DBAND_STAMP).

    python tools/band_stamps.py build [-D...]     # here (hipcc cross-compiles): tools/libtorbi_hip_bstamp.so
    python tools/band_stamps.py [frames]          # on the GPU box: 512 x frames x 1440, the pitch band, peaked rows
"""
import ctypes, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get('STAMP_LIB') or os.path.join(ROOT, 'tools', 'libtorbi_hip_bstamp.so')

if len(sys.argv) > 1 and sys.argv[1] == 'build':
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
                           '-fno-slp-vectorize', '-Wno-pass-failed', '-DBAND_STAMP', f'-I{ROOT}/include', '-o', LIB,
                           f'{ROOT}/torbi_amd/csrc/torbi_hip.hip'] + sys.argv[2:])
    sys.exit(0)

import numpy as np, torch
import torbi_amd._lib as _lib
_lib.LIBRARY = LIB
import torbi_amd
from torbi_amd import synth, viterbi
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, S = 512, 1440
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev).manual_seed(7)
logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
obs = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
band = torch.from_numpy(synth.banded_transition(S, 87.2)).to(dev)
init = torch.full((S,), math.log(1.0 / S), device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
prof = []
for _ in range(3):
    viterbi.decode(obs, frames, band, init, path='band', _profile=prof)
torch.cuda.synchronize()
lib = _lib.load()
KW, KP, nwg = 12, 12, 256
buf = (ctypes.c_ulonglong * (nwg * KW * KP))()
lib.torbi_hip_debug_band_phases.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.torbi_hip_debug_band_phases(buf, nwg * KW * KP)
acc = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, KW, KP).astype(np.float64)
steps = T - 1
names = ['own-row dquads, first run', 'own-row dquads, second run', 'wait for the halo granules', 'halo -> window + barrier',
         'halo dquads', 'merge through M', 'barrier (all waves merged)', 'finish: M -> window, history, exchange', 'barrier (window complete)']
tot = acc[:, :, :9].sum(axis=2).mean()
print(f'rc {rc}; route {viterbi.ROUTES[int(prof[3])]}; forward {prof[0]:.3f} ms for {B} x {T}; {tot / steps:.0f} ticks per timestep per wave '
      f'({prof[0] * 1e3 / steps:.2f} us per timestep -> {tot / steps / (prof[0] * 1e3 / steps) / 1e3:.2f} GHz)')
for i, name in enumerate(names):
    v = acc[:, :, i].mean() / steps
    print(f'{name:42s} {v:8.0f} ticks/step  {100 * v * steps / tot:5.1f} %   (min / max over waves {acc[:, :, i].min() / steps:.0f} / {acc[:, :, i].max() / steps:.0f})')
print(f'failed polls per wave and timestep: {acc[:, :, 9].mean() / steps:.2f}')
