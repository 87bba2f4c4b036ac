"""Up to 64 states: one wavefront per sequence in one launch (csrc/small_states.hpp) against the per-timestep trellis
kernels (DENSE named below 64 states = the generic route) and the other routes that cover 64 states.
    python tools/small_states_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torbi_amd
from torbi_amd import viterbi, synth

dev = torch.device('cuda:0')


def timed(fn, n=7):
    fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


SHAPES = [(1, 500, 3), (1, 500, 40), (1, 5000, 64), (16, 500, 40), (512, 500, 32), (512, 500, 64), (4096, 500, 64),
          (32768, 200, 8)]
if len(sys.argv) > 1:          # e.g.  1x500x128 16x500x128
    SHAPES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (B, T, S) in SHAPES:
    obs = torch.randn(B, T, S, device=dev).log_softmax(-1)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    _, trans, init = synth.problem(1, 1, S, seed=3)
    trans, init = torch.as_tensor(trans).to(dev), torch.as_tensor(init).to(dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    line = f'{B} x {T} x {S}:'
    want, seen = None, set()
    for path in os.environ.get('PROBE_PATHS', 'auto,dense,pruned,resident,cluster,held').split(','):   # (PROBE_PATHS=auto: profiling)
        route = viterbi.forward_path(B, S, path=path)
        if path != 'auto' and route in seen:
            continue
        seen.add(route)
        if route == 'generic' and B * T > 2_000_000:
            continue
        got = viterbi.decode(obs, frames, trans, init, workspace=ws, path=path)
        want = got if want is None else want
        assert torch.equal(got, want), (path, route)
        ms = timed(lambda: viterbi.decode(obs, frames, trans, init, workspace=ws, path=path))
        line += f'  {route} {ms:.3f} ms ({B * T / ms / 1e3:.2f} M timesteps/s)'
    print(line, flush=True)
