"""A handful of sequences: what AUTO takes (held / generic / rows / cluster) against the cluster form and the generic kernels
named.  python tools/few_items_probe.py [S] [T]"""
import sys, time, torch
sys.path.insert(0,'/root/repo')
import torbi_amd
from torbi_amd import viterbi, synth
dev=torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
for B in (2,3,4,5,6,8,12,16,17,24):
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    line=f'B={B}:'
    ref=None
    for path in ('auto','cluster','dense','pruned'):
        got=torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path)
        ref = got if ref is None else ref
        ts=[]
        for _ in range(5):
            torch.cuda.synchronize(); t0=time.perf_counter(); torbi_amd.decode(obs, frames, trans, init, workspace=ws, path=path); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
        line+=f'  {path} [{viterbi.forward_path(B,S,path=path)}] {sorted(ts)[2]*1e3:.2f} ms eq {torch.equal(got,ref)}'
    print(line, flush=True)
