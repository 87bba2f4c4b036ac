#!/bin/bash
# GPU box: the 4096-file job with and without serialised native reads, alternating on ONE box (files written once)
cd "$(dirname "$0")/.."
python - <<'PY'
import os, shutil, sys, tempfile, time, subprocess
sys.path.insert(0, '.')
import torch
from torbi_amd import synth
files, S = 4096, 1440
lengths = synth.lengths(files, 100, 900).tolist()
folder = tempfile.mkdtemp(prefix='torbi_ab_', dir='/dev/shm')
try:
    gen = torch.Generator().manual_seed(1)
    block = torch.rand(900, S, generator=gen).log_softmax(-1)
    for k, n in enumerate(lengths):
        torch.save(torch.roll(block, k, dims=0)[:n].clone(), os.path.join(folder, f'in{k}.pt'))
    torch.save(torch.rand(S, S, generator=gen).softmax(-1), os.path.join(folder, 'transition.pt'))
    code = '''
import os, sys, time
sys.path.insert(0, ".")
import torch, torbi_amd
from torbi_amd import synth
folder, files = sys.argv[1], 4096
lengths = synth.lengths(files, 100, 900).tolist()
ins = [os.path.join(folder, f"in{k}.pt") for k in range(files)]
outs = [os.path.join(folder, f"out{k}.pt") for k in range(files)]
tf = os.path.join(folder, "transition.pt")
res = []
for attempt in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, lengths=lengths, gpu=0, num_workers=32)
    torch.cuda.synchronize(); res.append(time.perf_counter() - t0)
print(os.environ.get("TORBI_SERIAL_READS"), " ".join(f"{x:.3f}" for x in res))
'''
    for rep in range(2):
        for mode in ('1', '0'):
            out = subprocess.run([sys.executable, '-c', code, folder], env=dict(os.environ, TORBI_SERIAL_READS=mode), capture_output=True, text=True)
            print('serial reads', out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
finally:
    shutil.rmtree(folder, ignore_errors=True)
PY
