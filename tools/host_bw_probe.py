"""Host side of the many-file job, piece by piece (GPU box): the native reader (page cache -> pinned batch rows), the H2D
copy, both at once, by reader thread count and CPU affinity (NUMA node of the GPU or not).
    python tools/host_bw_probe.py [files=512]"""
import ctypes, glob, os, shutil, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torbi_amd import _lib, fastio, synth

files = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S, T = 1440, 900
dev = torch.device('cuda:0')


def nodes():
    out = {}
    for path in sorted(glob.glob('/sys/devices/system/node/node[0-9]*')):
        text = open(os.path.join(path, 'cpulist')).read().strip()
        cpus = []
        for part in text.split(','):
            lo, _, hi = part.partition('-')
            cpus += list(range(int(lo), int(hi or lo) + 1))
        out[int(path.rsplit('node', 1)[1])] = cpus
    return out


topo = nodes()
gpu_nodes = [open(f).read().strip() for f in glob.glob('/sys/class/drm/card*/device/numa_node')]
print('numa nodes:', {k: f'{len(v)} cpus ({v[0]}..{v[-1]})' for k, v in topo.items()}, 'gpu numa_node:', gpu_nodes,
      'affinity now:', len(os.sched_getaffinity(0)), flush=True)

folder = tempfile.mkdtemp(prefix='torbi_bw_', dir='/dev/shm')
try:
    block = torch.rand(T, S).log_softmax(-1)
    names = []
    for k in range(files):
        f = os.path.join(folder, f'in{k}.pt'); torch.save(block.clone(), f); names.append(f)
    opened = [fastio._open_payload(f) for f in names]
    row_bytes = 4 * T * S
    gb = files * row_bytes / 1e9
    fds = np.array([e[0] for e in opened], dtype=np.int32)
    offsets = np.array([e[3] for e in opened], dtype=np.int64)
    sizes = np.full(files, row_bytes, dtype=np.int64)
    zeros = np.zeros(files, dtype=np.int64)
    read_rows, _ = _lib.host_io(True)

    def pinned():
        return torch.empty((files, T, S), dtype=torch.float32, pin_memory=True)

    def read_into(buf, threads):
        rows = buf.data_ptr() + row_bytes * np.arange(files, dtype=np.int64)
        err = ctypes.c_int(0)
        t0 = time.perf_counter()
        rc = read_rows(fds.ctypes.data, offsets.ctypes.data, sizes.ctypes.data, rows.ctypes.data, zeros.ctypes.data, files, threads,
                       ctypes.byref(err))
        assert rc == 0
        return time.perf_counter() - t0

    device = torch.empty((files, T, S), dtype=torch.float32, device=dev)
    copy_stream = torch.cuda.Stream(device=dev)

    def h2d(buf, reps=1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(copy_stream):
            for _ in range(reps):
                device.copy_(buf, non_blocking=True)
        copy_stream.synchronize()
        return (time.perf_counter() - t0) / reps

    def both(a, b, threads):
        """reader fills `a` while `b` is copied to the device"""
        torch.cuda.synchronize()
        done = {}
        def copier():
            done['h2d'] = h2d(b, reps=2)
        th = threading.Thread(target=copier); t0 = time.perf_counter(); th.start()
        done['read'] = min(read_into(a, threads), 1e9)
        done['read2'] = read_into(a, threads)
        th.join()
        return done, time.perf_counter() - t0

    def suite(label):
        a, b = pinned(), pinned()
        read_into(a, 32); read_into(b, 32)             # first touch
        for threads in (8, 16, 32, 64, 128):
            dt = min(read_into(a, threads) for _ in range(3))
            print(f'{label} reader alone  threads={threads:3d}: {gb / dt:6.1f} GB/s', flush=True)
        print(f'{label} H2D alone: {gb / h2d(b, reps=3):6.1f} GB/s', flush=True)
        for threads in (16, 32, 64, 128):
            d, wall = both(a, b, threads)
            print(f'{label} together threads={threads:3d}: reader {gb / d["read"]:6.1f} / {gb / d["read2"]:6.1f} GB/s, H2D {gb / d["h2d"]:6.1f} GB/s '
                  f'(2 reads + 2 copies in {wall:.3f} s = {4 * gb / wall:6.1f} GB/s moved)', flush=True)
        del a, b

    suite('all cpus      ')
    everything = os.sched_getaffinity(0)
    for node, cpus in topo.items():
        os.sched_setaffinity(0, set(cpus) & everything or everything)
        suite(f'node {node} cpus   ')
    os.sched_setaffinity(0, everything)
finally:
    shutil.rmtree(folder, ignore_errors=True)
