import sys, time, torch
sys.path.insert(0, '.')
import torbi_amd
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S = 512, 500, 1440
obs = viterbi.fill_synthetic((B, T, S), 1, device=dev); trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
n = viterbi.workspace_bytes(B, T, S)
ws = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def run(K, two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for k in range(K):
        if two:
            with torch.cuda.stream(streams[k % 2]):
                outs.append(torbi_amd.decode(obs, frames, trans, init, workspace=ws[k % 2]))
        else:
            outs.append(torbi_amd.decode(obs, frames, trans, init, workspace=ws[0]))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt / K * 1e3, outs
run(2, False); run(2, True)
a, o1 = run(10, False); b, o2 = run(10, True)
print('one stream ms/decode', a, ' two streams', b, 'equal', all(torch.equal(x, y) for x, y in zip(o1, o2)))
