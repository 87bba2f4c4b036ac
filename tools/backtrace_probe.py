"""Backtrace of a launch group: posterior rows staged in the LDS against posteriors gathered where the lists point
(TORBI_HIP_BACKTRACE_GATHER=0/1; read once per process, so one process per setting).
    TORBI_HIP_BACKTRACE_GATHER=1 python tools/backtrace_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torbi_amd import viterbi, synth

T, S, B = 500, 1440, 512
dev = torch.device('cuda:0')
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
out = []
ref = None
for n in (1, 2, 4, 8):
    obs = [viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=k, device=dev) for k in range(n)]
    frames = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]
    ws = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
    best_f, best_b = 1e9, 1e9
    for rep in range(3):
        prof = []
        got = viterbi.decode_batches(obs, frames, trans, init, workspaces=ws, _profile=prof)
        torch.cuda.synchronize()
        best_f, best_b = min(best_f, prof[0]), min(best_b, prof[1])
    check = int(sum(int(g.sum()) for g in got))
    out.append(f'{n} x {B}: forward {best_f:6.2f} ms, backtrace {best_b:5.2f} ms, checksum {check}')
print(f'gather = {os.environ.get("TORBI_HIP_BACKTRACE_GATHER", "auto")}: ' + ' | '.join(out))
