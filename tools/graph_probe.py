"""Does a decode capture into a HIP graph, and what does a replay save?  torbi_amd.decode with a caller-owned workspace under
torch.cuda.graph: shapes of one launch (one sequence, small state counts) and of the cluster form; replays on NEW observations
compared with eager decodes of the same.  (GPU box)    python tools/graph_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth

dev = torch.device('cuda:0')


def run(B, T, S, band=False):
    init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
    if band:
        trans = torch.from_numpy(synth.banded_transition(S, 87.2)).to(dev)
    else:
        trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=1, device=dev)
    other = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=2, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    for _ in range(3):                       # eager first: routing decided, preparation done, the look at the matrix taken
        want = torbi_amd.decode(obs, frames, trans, init, workspace=ws)
    prof = []
    want_other = torbi_amd.decode(other, frames, trans, init, workspace=ws, _profile=prof).clone()
    route = viterbi.ROUTES[int(prof[3])]
    want = want.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        torbi_amd.decode(obs, frames, trans, init, workspace=ws)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    static = obs.clone()
    side = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            torbi_amd.decode(static, frames, trans, init, workspace=ws)
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                out = torbi_amd.decode(static, frames, trans, init, workspace=ws)
    except Exception as exc:                 # noqa: BLE001
        print(f'{B} x {T} x {S}{" band" if band else ""}: capture failed: {type(exc).__name__}: {str(exc)[:300]}', flush=True)
        return
    graph.replay()
    torch.cuda.synchronize()
    same = torch.equal(out, want)
    static.copy_(other)
    graph.replay()
    torch.cuda.synchronize()
    same_other = torch.equal(out, want_other)
    t0 = time.perf_counter()
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / 20
    print(f'{B} x {T} x {S}{" band" if band else ""}: route {route}; eager {eager * 1e3:.3f} ms, replay {replay * 1e3:.3f} ms; '
          f'replay == eager: {same}, on new observations: {same_other}', flush=True)


if len(sys.argv) > 1:                       # one shape per process: a capture that fails leaves the process's HIP state unusable
    B, T, S, band = (int(a) for a in sys.argv[1:5])
    run(B, T, S, band=bool(band))
else:
    import subprocess
    for shape in ((1, 500, 1440, 0), (8, 500, 1440, 0), (512, 500, 64, 0), (512, 500, 256, 0), (512, 200, 1440, 0), (2048, 100, 1440, 0),
                  (512, 200, 1440, 1), (2560, 100, 1440, 1), (128, 300, 4096, 0)):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), *map(str, shape)], capture_output=True, text=True)
        lines = [l for l in (out.stdout + out.stderr).splitlines() if ' x ' in l and ('route' in l or 'capture failed' in l)]
        print(lines[0] if lines else f'{shape}: no result: ' + (out.stdout + out.stderr)[-400:], flush=True)
