"""Up to 64 states: the byte-backpointer wavefront kernel against its value-only form (csrc/small_states.hpp,
TORBI_HIP_SMALL_VALUE=0|1), in child processes (the switch is read once).  python tools/small_value_probe.py"""
import os, subprocess, sys
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else %r)
import torbi_amd
from torbi_amd import viterbi, synth
import numpy as np
dev = torch.device('cuda:0')
for (B, T, S) in [(512, 500, 3), (512, 500, 40), (512, 500, 64), (2048, 500, 64), (4096, 500, 64), (4096, 500, 40), (8192, 500, 16), (32768, 200, 8)]:
    obs = torch.randn(B, T, S, device=dev).log_softmax(-1)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    _, trans, init = synth.problem(1, 1, S, seed=3)
    trans, init = torch.as_tensor(trans).to(dev), torch.as_tensor(init).to(dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    got = viterbi.decode(obs, frames, trans, init, workspace=ws)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); viterbi.decode(obs, frames, trans, init, workspace=ws); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ms = sorted(ts)[3] * 1e3
    import hashlib
    print(f'{B} x {T} x {S}: {viterbi.last_forward_kernel():40s} {ms:8.3f} ms  {B * T * (8 * S + 8) / ms / 1e9 / 8000 * 100:5.1f} %% of HBM roofline  sha {hashlib.sha1(got.cpu().numpy().tobytes()).hexdigest()[:10]}', flush=True)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for form in ('0', '1'):
    print(f'== TORBI_HIP_SMALL_VALUE={form}', flush=True)
    subprocess.run([sys.executable, '-c', CHILD % root], env=dict(os.environ, TORBI_HIP_SMALL_VALUE=form))
