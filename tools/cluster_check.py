"""Cluster form of the time-resident kernel against the oracle (small shapes) and against the per-timestep path
(timing, headline batch).  GPU box:  timeout 300 python tools/cluster_check.py [quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import oracle
import torbi_amd
from torbi_amd import viterbi, synth

dev = torch.device('cuda:0')


def gave_up(ws, B, T, S):
    stats = viterbi.scan_stats(ws, B, T, S, path='cluster')
    return int(stats.cpu()[127])


bad = 0
for (B, T, S) in [(16, 6, 64), (17, 9, 96), (40, 12, 360), (33, 20, 1440), (100, 7, 130), (512, 5, 1440), (64, 9, 2048),
                  (1, 30, 1440), (3, 12, 1442), (250, 9, 720), (40, 6, 4096), (3, 5, 2052), (130, 5, 3000), (9, 7, 4094)]:
    obs, trans, init = synth.problem(B, T, S, seed=B + T)
    frames = np.clip(synth.lengths(B, 1, T, seed=S), 1, T)
    frames[0] = T
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    d = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    for path in ('cluster', 'resident'):
        for rep in range(3):
            got = torbi_amd.decode(*d, workspace=ws, path=path)
            torch.cuda.synchronize()
            ok = np.array_equal(got.cpu().numpy(), want)
            g = gave_up(ws, B, T, S)
            if not ok or g:
                bad += 1
                print(f'MISMATCH {(B, T, S)} {path} rep {rep}: equal {ok}, gave up {g}, route {viterbi.forward_path(B, S, path)}', flush=True)
                break
        else:
            print(f'ok {(B, T, S)} route {viterbi.forward_path(B, S, path)}', flush=True)
print('small shapes:', 'ALL OK' if not bad else f'{bad} BAD', flush=True)
if bad or (len(sys.argv) > 1 and sys.argv[1] == 'quick'):
    sys.exit(1 if bad else 0)

B, T, S = 512, 200, 1440
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
ref = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path='pruned')
for path in ('pruned', 'cluster', 'resident'):
    for rep in range(3):
        prof = []
        got = torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path, _profile=prof)
    torch.cuda.synchronize()
    print(f'{path:9s} B={B} T={T} S={S}: forward {prof[0]:.3f} ms (prep {prof[4]:.3f}) = {1e3 * (prof[0] - prof[4]) / (T - 1):.2f} us/step, '
          f'backtrace {prof[1]:.3f} ms, route {int(prof[3])}, equal {torch.equal(got, ref)}, gave up {gave_up(ws, B, T, S) if path == "cluster" else "-"}',
          flush=True)

# launch groups that do not fill the chip: clusters of 2 / 4 workgroups per tile against whole tiles on some of the CUs
for n in (4, 2, 3, 5):
    obs_n = [viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=k, device=dev) for k in range(n)]
    ws_n = [torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(n)]
    want = None
    for path in ('resident', 'cluster'):
        for rep in range(3):
            prof = []
            got = viterbi.decode_batches(obs_n, [frames] * n, trans, init, workspaces=ws_n, path=path, _profile=prof)
        torch.cuda.synchronize()
        if want is None:
            want = got
        same = all(torch.equal(a, b) for a, b in zip(got, want))
        print(f'{n} batches {path:9s}: forward {prof[0]:.3f} ms (prep {prof[4]:.3f}) = {1e3 * (prof[0] - prof[4]) / (T - 1):.2f} us/step, '
              f'backtrace {prof[1]:.3f} ms, route {int(prof[3])}, same indices {same}, gave up {gave_up(ws_n[0], B, T, S)}', flush=True)
