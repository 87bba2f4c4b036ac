"""Soak of the many-file job's host side (reader threads, ring of pinned chunks, copy / preparation streams, saver threads): the
same ragged job again and again with random ring sizes, reader counts and batch sizes, every output file compared with the
files of a first run that staged whole batches (RING_CHUNKS = 0), itself checked against the dense route.  (GPU box)
    python tools/file_job_soak.py [seconds] [files] [states]"""
import os, random, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torbi_amd
from torbi_amd import core, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
files = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
lengths = synth.lengths(files, 20, 300).tolist()
folder = tempfile.mkdtemp(prefix='torbi_soak_', dir='/dev/shm')
rng = random.Random(5)
try:
    gen = torch.Generator().manual_seed(1)
    probs = bool(os.environ.get('PROBS'))          # files of probabilities, log_probs=False (the reference's default call)
    block = torch.rand(300, S, generator=gen).softmax(-1) if probs else torch.rand(300, S, generator=gen).log_softmax(-1)
    ins = []
    for k, n in enumerate(lengths):
        f = os.path.join(folder, f'in{k}.pt')
        torch.save(torch.roll(block, 7 * k, dims=0)[:n].roll(k, dims=1).clone(), f)
        ins.append(f)
    tf = os.path.join(folder, 'transition.pt')
    torch.save(torch.rand(S, S, generator=gen).mul_(4.0).softmax(-1), tf)
    first = [os.path.join(folder, f'first{k}.pt') for k in range(files)]
    core.RING_CHUNKS = 0
    torbi_amd.from_files_to_files(ins, first, transition_file=tf, log_probs=not probs, gpu=0)
    want = [torch.load(f) for f in first]
    # the first run against single decodes of some files on the dense route
    trans = (torch.log(torch.load(tf)) if probs else torch.log(torch.load(tf) + torch.finfo(torch.float32).tiny)).cuda()
    init = torch.full((S,), float(torch.log(torch.tensor(1.0 / S) + torch.finfo(torch.float32).tiny))).cuda()
    for k in range(0, files, 97):
        x = torch.log(torch.load(ins[k]).cuda()) if probs else torch.load(ins[k]).cuda()
        x = torch.log(torch.exp(x) + torch.finfo(torch.float32).tiny)[None]
        got = torbi_amd.decode(x, torch.tensor([lengths[k]], dtype=torch.int32, device='cuda'), trans, init, path='dense')[0].cpu()
        assert torch.equal(got[:lengths[k]], want[k]), k
    t0, rounds, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        core.RING_CHUNKS = rng.choice([1, 2, 3, 4, 6])
        core.RING_CHUNK_BYTES = rng.choice([1 << 12, 1 << 16, 1 << 20, 1 << 22, 1 << 28])
        core.BATCH_SIZE = rng.choice([512, 512, 200, 64])
        core.KEEP_JOB_MEMORY = rng.random() < 0.5
        outs = [os.path.join(folder, f'out{k}.pt') for k in range(files)]
        torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=not probs, gpu=0, num_workers=rng.choice([2, 8, 16, 32]),
                                      lengths=lengths if rng.random() < 0.5 else None)
        for k, f in enumerate(outs):
            if not torch.equal(torch.load(f), want[k]):
                bad += 1
                print(f'round {rounds}: file {k} differs (chunks {core.RING_CHUNKS}, bytes {core.RING_CHUNK_BYTES}, batch {core.BATCH_SIZE})', flush=True)
            os.remove(f)
        rounds += 1
    print(f'{rounds} jobs x {files} files in {time.time() - t0:.0f} s: {bad} differing files', flush=True)
    sys.exit(1 if bad else 0)
finally:
    shutil.rmtree(folder, ignore_errors=True)
