"""Determinism soak (GPU box): the headline decode repeated through the two-stream pipeline, every result compared with
the first; then ragged lengths.   python tools/soak.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import synth, viterbi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, T, S = 512, 500, 1440
dev = torch.device('cuda:0')
obs = viterbi.fill_synthetic((B, T, S), 1, device=dev); trans = viterbi.fill_synthetic((S, S), 2, device=dev)
init = viterbi.fill_synthetic((S,), 3, device=dev)
for name, frames in (('full', torch.full((B,), T, dtype=torch.int32, device=dev)),
                     ('ragged', torch.as_tensor(synth.lengths(B, 1, T, seed=4)).to(dev))):
    pipe = torbi_amd.DecodePipeline(dev)
    outs = [pipe.decode(obs, frames, trans, init) for _ in range(n)]
    pipe.synchronize()
    bad = sum(int(not torch.equal(o, outs[0])) for o in outs[1:])
    for path in ('dense', 'resident'):
        viterbi.set_forward_path(path)
        bad += int(not torch.equal(torbi_amd.decode(obs, frames, trans, init), outs[0]))
    viterbi.set_forward_path('auto')
    print(name, n, 'pipelined repeats + forced dense/pruned:', bad, 'differences')
