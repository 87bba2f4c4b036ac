"""cProfile of the calling thread of torbi_amd.from_files_to_files on a ragged many-file job in /dev/shm.
python tools/file_job_profile.py [files] [threads]"""
import cProfile, os, pstats, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import synth

files = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = 1440
lengths = synth.lengths(files, 100, 900).tolist()
folder = tempfile.mkdtemp(prefix='torbi_job_', dir='/dev/shm')
try:
    gen = torch.Generator().manual_seed(1)
    probs = bool(os.environ.get('PROBS'))              # files of probabilities, log_probs=False: the reference's default call
    block = torch.rand(900, S, generator=gen).softmax(-1) if probs else torch.rand(900, S, generator=gen).log_softmax(-1)
    ins, outs = [], []
    for k, n in enumerate(lengths):
        f = os.path.join(folder, f'in{k}.pt'); torch.save(torch.roll(block, k, dims=0)[:n].clone(), f)
        ins.append(f); outs.append(os.path.join(folder, f'out{k}.pt'))
    tf = os.path.join(folder, 'transition.pt'); torch.save(torch.rand(S, S, generator=gen).softmax(-1), tf)
    if os.environ.get('PREREAD'):          # every input file read once by plain Python before the first job (is the cold
        t0 = time.perf_counter()           # job slow because the FILES are new, or because the process is?)
        n = 0
        for f in ins:
            with open(f, 'rb') as fh:
                n += len(fh.read())
        print(f'pre-read {n / 1e9:.1f} GB in {time.perf_counter() - t0:.2f} s', flush=True)
    if os.environ.get('PREPIN'):           # ... or because the pinned memory is?
        t0 = time.perf_counter()
        warm = [torch.empty(int(2.8e9), dtype=torch.uint8).pin_memory() for _ in range(4)]
        for w in warm:
            w.zero_()
        del warm
        print(f'pinned and touched 4 x 2.8 GB in {time.perf_counter() - t0:.2f} s', flush=True)
    for attempt in range(2):
        torch.cuda.synchronize()
        prof = cProfile.Profile()
        t0 = time.perf_counter()
        prof.enable()
        torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=not probs, lengths=lengths, gpu=0, num_workers=threads)
        torch.cuda.synchronize()
        prof.disable()
        dt = time.perf_counter() - t0
        print(f'run {attempt}: {dt:.2f} s, {sum(lengths) / dt / 1e6:.2f} M frames/s', flush=True)
        from torbi_amd import slabs, fastio
        for rec in getattr(fastio, 'LAST_TIMINGS', None) or []:
            print(f'  batch: open+headers {rec[0] * 1e3:6.1f} ms, slab {rec[1] * 1e3:6.1f} ms, native read {rec[2] * 1e3:6.1f} ms ({rec[3] / rec[2] / 1e9:5.1f} GB/s)')
        for name, p in slabs._pools.items():
            print(f'  slab pool {name}: {len(p._free)} free, {len(p._busy)} busy, {p.held_bytes() / 1e9:.1f} GB', flush=True)
        pstats.Stats(prof).sort_stats('cumulative').print_stats(28)
finally:
    shutil.rmtree(folder, ignore_errors=True)
