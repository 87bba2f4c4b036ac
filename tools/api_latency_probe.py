"""What a caller of the reference's API sees for ONE sequence (BASELINE configs[1]): torbi_amd.from_probabilities on
host and device tensors, against the operator alone.   python tools/api_latency_probe.py [S] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
probs = torch.rand(1, T, S, generator=g).softmax(-1)
trans = torch.rand(S, S, generator=g).softmax(-1)
init = torch.rand(S, generator=g).softmax(-1)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms, a = timed(lambda: torbi_amd.from_probabilities(probs, transition=trans, initial=init, gpu=0))
print(f'from_probabilities, host tensors in, host indices out: {ms:7.3f} ms')
d_probs, d_trans, d_init = probs.to(dev), trans.to(dev), init.to(dev)
ms, b = timed(lambda: torbi_amd.from_probabilities(d_probs, transition=d_trans, initial=d_init, gpu=0))
print(f'from_probabilities, device tensors in:                 {ms:7.3f} ms')
obs, ltrans, linit = torch.log(d_probs), torch.log(d_trans), torch.log(d_init)
frames = torch.full((1,), T, dtype=torch.int32, device=dev)
ms, c = timed(lambda: viterbi.decode(obs, frames, ltrans, linit))
print(f'decode (log-space device tensors, workspace from the caching allocator): {ms:7.3f} ms')
space = torch.empty(viterbi.workspace_bytes(1, T, S), dtype=torch.uint8, device=dev)
ms, d = timed(lambda: viterbi.decode(obs, frames, ltrans, linit, workspace=space))
print(f'decode with a caller-owned workspace:                  {ms:7.3f} ms')
print('same indices:', bool(torch.equal(a.to(dev), b.to(dev))))

# where the host-tensor call spends its time
def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
ms, lg = t(lambda: torch.log(probs)); print(f'  torch.log on the host ({torch.get_num_threads()} threads): {ms:7.3f} ms')
torch.set_num_threads(1)
ms, lg = t(lambda: torch.log(probs)); print(f'  torch.log on the host (1 thread):  {ms:7.3f} ms')
ms, dv = t(lambda: lg.to(dev)); print(f'  pageable H2D of {lg.numel() * 4 / 1e6:.1f} MB:        {ms:7.3f} ms')
pin = lg.pin_memory()
ms, dv = t(lambda: pin.to(dev, non_blocking=True)); print(f'  pinned H2D:                        {ms:7.3f} ms')
ms, _ = t(lambda: torbi_amd.epsilon_clamp_(dv) if hasattr(torbi_amd, 'epsilon_clamp_') else None); print(f'  epsilon clamp:                     {ms:7.3f} ms')
idx = viterbi.decode(obs, frames, ltrans, linit)
ms, _ = t(lambda: idx.cpu()); print(f'  indices to the host:               {ms:7.3f} ms')
ms, _ = t(lambda: torbi_amd.from_probabilities(probs, transition=trans, initial=init, gpu=0)); print(f'from_probabilities with 1 host thread: {ms:7.3f} ms')
