"""What the reference API's call patterns cost on one 512 x 500 x 1440 batch: where the observation lives (device, pinned,
pageable), what it holds (log-probabilities / probabilities), the model (dense matrix kept / made anew per call / none), the
operator entry (torch.ops.torbi.viterbi_decode).  Looks for call patterns that are far off what their bytes explain.
(GPU box)    python tools/api_matrix_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi, synth, torch_op
torch_op.register()

dev = torch.device('cuda:0')
B, T, S = 512, 500, 1440
logp = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=1, device=dev).log_softmax(-1)
prob = logp.exp()
trans_p = torch.rand(S, S, device=dev).mul_(4.0).softmax(-1)
trans_l = trans_p.log()
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
host_l, host_p = logp.cpu(), prob.cpu()
pin_l, pin_p = host_l.pin_memory(), host_p.pin_memory()
trans_p_host, trans_l_host = trans_p.cpu(), trans_l.cpu()


def timed(name, fn, n=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{name:86s} {dt * 1e3:9.2f} ms  {B * T / dt / 1e6:7.2f} M timesteps/s', flush=True)
    return out


ref = timed('decode(device log-probs, workspace kept)', lambda: torbi_amd.decode(logp, frames, trans_l, None if False else torch.full((S,), -7.27, device=dev)))
calls = [
    ('from_probabilities(device log-probs, device matrix kept, log_probs=True)', lambda: torbi_amd.from_probabilities(logp.clone(), frames, trans_l, log_probs=True, gpu=0)),
    ('from_probabilities(device probabilities, device matrix of probabilities kept)', lambda: torbi_amd.from_probabilities(prob, frames, trans_p, gpu=0)),
    ('from_probabilities(device probabilities, matrix of probabilities made anew per call)', lambda: torbi_amd.from_probabilities(prob, frames, trans_p.clone(), gpu=0)),
    ('from_probabilities(device probabilities, no model: every default)', lambda: torbi_amd.from_probabilities(prob, gpu=0)),
    ('from_probabilities(device log-probs, no model, log_probs=True)', lambda: torbi_amd.from_probabilities(logp.clone(), log_probs=True, gpu=0)),
    ('from_probabilities(pinned log-probs, host matrix kept, log_probs=True)', lambda: torbi_amd.from_probabilities(pin_l, frames.cpu(), trans_l_host, log_probs=True, gpu=0)),
    ('from_probabilities(pageable log-probs, host matrix kept, log_probs=True)', lambda: torbi_amd.from_probabilities(host_l, frames.cpu(), trans_l_host, log_probs=True, gpu=0)),
    ('from_probabilities(pinned probabilities, host matrix of probabilities kept)', lambda: torbi_amd.from_probabilities(pin_p, frames.cpu(), trans_p_host, gpu=0)),
    ('from_probabilities(pageable probabilities, host matrix of probabilities kept)', lambda: torbi_amd.from_probabilities(host_p, frames.cpu(), trans_p_host, gpu=0)),
    ('torch.ops.torbi.viterbi_decode(device tensors)', lambda: torch.ops.torbi.viterbi_decode(logp, frames, trans_l, torch.full((S,), -7.27, device=dev))),
]
for name, fn in calls:
    timed(name, fn)
