import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import torbi_amd
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
for (B, T, S) in [(4096, 500, 64), (512, 500, 40), (4096, 500, 40)]:
    obs = torch.randn(B, T, S, device=dev).log_softmax(-1)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    _, trans, init = synth.problem(1, 1, S, seed=3)
    trans, init = torch.as_tensor(trans).to(dev), torch.as_tensor(init).to(dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    got = viterbi.decode(obs, frames, trans, init, workspace=ws)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); viterbi.decode(obs, frames, trans, init, workspace=ws); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'{B} x {T} x {S}: {viterbi.last_forward_kernel():40s} {sorted(ts)[3]*1e3:8.3f} ms', flush=True)
