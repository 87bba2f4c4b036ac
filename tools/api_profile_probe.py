import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, torbi_amd
S, T = 1440, 500
g = torch.Generator().manual_seed(0)
probs = torch.rand(1, T, S, generator=g).softmax(-1)
trans = torch.rand(S, S, generator=g).softmax(-1)
init = torch.rand(S, generator=g).softmax(-1)
f = lambda: torbi_amd.from_probabilities(probs, transition=trans, initial=init, gpu=0)
for _ in range(5): f()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): f()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
for k in (1, 2, 4, 8, 16, 32, 64, 128):
    torch.set_num_threads(k)
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(20): f()
    print(k, 'threads:', round((time.perf_counter() - t0) / 20 * 1e3, 3), 'ms')
