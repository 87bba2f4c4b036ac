"""The CPU twin (torbi_amd.decode_cpu) across thread counts: python tools/cpu_twin_probe.py [items] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torbi_amd
from torbi_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
S = 1440
obs, trans, init = synth.problem(B, T, S, seed=1)
args = [torch.as_tensor(x) for x in (obs, np.full(B, T, np.int32), trans, init)]
ref = None
for nt in (1, 8, 32, 64, 128, None):
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); got = torbi_amd.decode_cpu(*args, num_threads=nt); best = min(best, time.perf_counter() - t0)
    ref = got if ref is None else ref
    print(f'{B} x {T} x {S}, threads {nt}: {B * T / best:9.0f} timesteps/s  equal {bool(torch.equal(got, ref))}', flush=True)
