import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, numpy as np, torbi_amd
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
for (B, T, S) in [(512, 2000, 1440), (512, 200, 1440), (64, 500, 1440)]:
    obs = torch.randn(B, T, S, device=dev).log_softmax(-1)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    _, trans, init = synth.problem(2, 2, S, seed=3)
    trans = torch.as_tensor(trans).to(dev); init = torch.as_tensor(init).to(dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    def timed(fn, n=7):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3
    a = timed(lambda: viterbi.decode(obs, frames, trans, init, workspace=ws))
    b = timed(lambda: viterbi.decode(obs, frames, trans, init, workspace=ws, reuse_preparation=True))
    c = timed(lambda: viterbi.decode(obs, frames, trans, init))
    t2 = trans.clone()
    d = timed(lambda: (t2.add_(0), viterbi.decode(obs, frames, t2, init)))   # new version every call: rebuilt every time
    print(f'{B}x{T}x{S}: workspace {a:.3f} ms, workspace+reuse {b:.3f}, no workspace (kept with the matrix) {c:.3f}, no workspace + matrix changed per call {d:.3f}')
