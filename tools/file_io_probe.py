"""Host side of the many-file job alone: how fast do batches of torch.save()d observation files reach pinned memory
(and the device)?  python tools/file_io_probe.py [files] [threads]      (GPU box; files live in /dev/shm)"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torbi_amd import synth, fastio, data

files = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = 1440
lengths = sorted(synth.lengths(files, 100, 900).tolist(), reverse=True)
folder = tempfile.mkdtemp(prefix='torbi_io_', dir='/dev/shm')
try:
    block = torch.rand(900, S).log_softmax(-1)
    names = []
    for k, n in enumerate(lengths):
        f = os.path.join(folder, f'in{k}.pt'); torch.save(block[:n].clone(), f); names.append(f)
    gb = sum(lengths) * S * 4 / 1e9
    dev = torch.device('cuda:0') if torch.cuda.is_available() else None

    def timed(label, batches, to_device):
        t0 = time.perf_counter(); last = t0; waits = []
        for observation, frames, _, _ in batches:
            now = time.perf_counter(); waits.append(now - last)
            if to_device and dev is not None:
                x = observation.to(dev, non_blocking=True)
            last = time.perf_counter()
        if dev is not None:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'{label:44s} {dt:6.2f} s  {gb / dt:6.2f} GB/s  {sum(lengths) / dt / 1e6:5.2f} M frames/s  waits {[round(w, 2) for w in waits]}', flush=True)

    t0 = time.perf_counter(); lay = [fastio.payload(f) for f in names[:512]]; print('payload() per file us', (time.perf_counter() - t0) / 512 * 1e6)
    for pin in (False, True, True):
        timed(f'FileBatches threads={threads} pin={pin}', fastio.FileBatches(names, 512, threads=threads, pin_memory=pin), False)
    timed(f'FileBatches threads={threads} pin=True + H2D', fastio.FileBatches(names, 512, threads=threads, pin_memory=True), True)
    timed(f'FileBatches threads={4 * threads} pin=True + H2D', fastio.FileBatches(names, 512, threads=4 * threads, pin_memory=True), True)
    timed(f'data.loader workers={threads} pin + H2D', data.loader(names, num_workers=threads), True)
    # pieces: pinned allocation, one big read
    for gbs in (1.0, 2.6):
        t0 = time.perf_counter(); x = torch.empty(int(gbs * 2**28), dtype=torch.float32, pin_memory=True); t1 = time.perf_counter()
        del x; t2 = time.perf_counter(); y = torch.empty(int(gbs * 2**28), dtype=torch.float32, pin_memory=True); t3 = time.perf_counter()
        print(f'pinned alloc {gbs} GiB: first {t1 - t0:.3f} s, again {t3 - t2:.3f} s'); del y
finally:
    shutil.rmtree(folder, ignore_errors=True)
