"""PCIe side of the file pipeline (GPU box): H2D rate of one 512 x 500 x 1440 fp32 batch from pageable and pinned
host memory, and batches/s of copy + decode with and without the DecodePipeline overlap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import viterbi
from torbi_amd.pipeline import DecodePipeline

B, T, S = 512, 500, 1440
dev = torch.device('cuda:0')
host = torch.empty((B, T, S), dtype=torch.float32).uniform_(-16, 0)
pinned = host.pin_memory()
trans = viterbi.fill_synthetic((S, S), 2, device=dev)
init = viterbi.fill_synthetic((S,), 3, device=dev)
frames = torch.full((B,), T, dtype=torch.int32, device=dev)
gb = host.numel() * 4 / 1e9
for name, src in (('pageable', host), ('pinned', pinned)):
    src.to(dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        src.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'H2D {name}: {dt * 1e3:.1f} ms per batch = {gb / dt:.1f} GB/s')

def serial(src, n=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        torbi_amd.decode(src.to(dev), frames, trans, init).cpu()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

def piped(src, n=6):
    pipe = DecodePipeline(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prev = None
    for _ in range(n):
        idx = pipe.decode(src.to(dev, non_blocking=True), frames, trans, init)
        if prev is not None:
            pipe.wait(prev); prev.cpu()
        prev = idx
    pipe.wait(prev); prev.cpu()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for name, src in (('pageable', host), ('pinned', pinned)):
    serial(src, 2); piped(src, 2)
    a, b = serial(src), piped(src)
    print(f'{name}: copy+decode+readback serial {a * 1e3:.1f} ms/batch ({B * T / a / 1e6:.1f} M timesteps/s), '
          f'pipelined {b * 1e3:.1f} ms/batch ({B * T / b / 1e6:.1f} M timesteps/s)')
