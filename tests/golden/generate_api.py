"""Generate tests/golden/golden_api.npz from the REFERENCE's own Python (SURVEY.md 8c, G6/G7 + chunking).

Run in the build container (where /root/reference exists and oracle/_ref has been built):

    python tests/golden/generate_api.py

The reference package cannot be imported as shipped (yapecs / torchutil / torchaudio are not installed and
its __init__ wants a `_C*.so` next to it), so this script assembles a throw-away copy under /tmp: symlinks to
the reference's .py files, the reference operator compiled by oracle/build.py as `_C.so`, and three stub
modules that provide exactly the names the package touches at import and on the file path
(`yapecs.configure/ArgumentParser`, `torchutil.time.context/iterator/metrics.Average`, empty `torchaudio`).
Nothing of the reference is copied into this repository; the .npz holds inputs and the outputs the reference
produced for them:

  collate/*      torbi.data.collate on ragged items                      (torbi/data/collate.py:9-33)
  files/*        torbi.from_files_to_files on small files, CPU operator   (torbi/core.py:310-368, 417-457)
  probs/*        torbi.from_probabilities with and without log_probs      (torbi/core.py:110-208)
  chunk/*        torbi.chunk entropy / cut points / pieces, and the chunked from_files_to_files
                 (torbi/chunk.py:12-85, data/dataset.py:22-23, collate.py:13-15,36-45, core.py:438-448)

Chunking has to be configured BEFORE `import torbi` (chunk()'s defaults are bound at import), so the script
re-runs itself as a child process with TORBI_STUB_MIN_CHUNK set; the stub `yapecs.configure` applies it.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden_api.npz')

STUBS = {
    'yapecs.py': '''
import argparse, os
def configure(name, defaults):
    value = os.environ.get('TORBI_STUB_MIN_CHUNK')
    if value:
        defaults.MIN_CHUNK_SIZE = int(value)
    value = os.environ.get('TORBI_STUB_BATCH_SIZE')
    if value:
        defaults.BATCH_SIZE = int(value)
ArgumentParser = argparse.ArgumentParser
''',
    'torchutil/__init__.py': '''
import contextlib
from . import time, metrics
def iterator(iterable, message=None, total=None):
    class Progress:
        def update(self, n=1): pass
        def close(self): pass
    return Progress()
def notify(*a, **k):
    return lambda function: function
''',
    'torchutil/time.py': '''
import contextlib
def context(name):
    return contextlib.nullcontext()
def reset(): pass
def results(): return {}
''',
    'torchutil/metrics.py': '''
class Average:
    def __init__(self): pass
''',
    'torchaudio.py': '',
}


def assemble(where):
    """/tmp copy of the reference package: symlinked sources + compiled operator + stubs."""
    sys.path.insert(0, ROOT)
    from oracle import build as oracle_build
    so = oracle_build.build_ref()
    assert so, 'oracle/_ref is not built (needs /root/reference)'
    pkg = os.path.join(where, 'torbi')
    for dirpath, _, files in os.walk(os.path.join(REFERENCE, 'torbi')):
        rel = os.path.relpath(dirpath, os.path.join(REFERENCE, 'torbi'))
        os.makedirs(os.path.join(pkg, rel), exist_ok=True)
        for f in files:
            if f.endswith(('.py', '.json')):
                os.symlink(os.path.join(dirpath, f), os.path.join(pkg, rel, f))
    shutil.copy(so, os.path.join(pkg, '_C.so'))
    for name, text in STUBS.items():
        path = os.path.join(where, name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            f.write(text)


def child(where, result_file):
    """Runs inside the assembled environment: import the real reference Python and record its outputs."""
    import torch
    sys.path.insert(0, where)
    import torbi
    torch.manual_seed(0)
    out = {}
    chunking = torbi.MIN_CHUNK_SIZE is not None
    tag = 'chunk' if chunking else 'plain'
    gen = torch.Generator().manual_seed(1234 if chunking else 4321)
    S = 24
    work = os.path.join(where, f'files_{tag}')
    os.makedirs(work)

    def near_certain(frames, states, certain):
        """log-probabilities; frames listed in `certain` put almost all mass on one state (low entropy)"""
        logits = torch.rand(frames, states, generator=gen) * 3.0
        for t in certain:
            logits[t, int(torch.randint(states, (1,), generator=gen))] += 40.0
        return logits.log_softmax(-1)

    if not chunking:
        # ---- collate on ragged items
        items = [(torch.rand(n, S, generator=gen), f'item{k}') for k, n in enumerate([7, 1, 12, 4])]
        observation, batch_frames, batch_chunks, names = torbi.data.collate(items)
        for k, (x, _) in enumerate(items):
            out[f'collate/item{k}'] = x.numpy()
        out['collate/observation'] = observation.numpy()
        out['collate/batch_frames'] = batch_frames.numpy()
        out['collate/batch_chunks'] = np.asarray(batch_chunks)
        out['collate/names'] = np.asarray(names)

        # ---- from_probabilities, probability and log-probability inputs (CPU operator)
        probs = torch.rand(3, 20, 50, generator=gen).softmax(-1)
        trans = torch.rand(50, 50, generator=gen).softmax(-1)
        init = torch.rand(50, generator=gen).softmax(-1)
        frames = torch.tensor([20, 13, 1], dtype=torch.int32)
        out['probs/observation'] = probs.numpy()
        out['probs/transition'] = trans.numpy()
        out['probs/initial'] = init.numpy()
        out['probs/batch_frames'] = frames.numpy()
        out['probs/indices'] = torbi.from_probabilities(probs.clone(), frames, trans, init, log_probs=False,
                                                        num_threads=1).numpy()
        out['probs/indices_defaults'] = torbi.from_probabilities(probs.clone(), num_threads=1).numpy()
        out['probs/indices_log'] = torbi.from_probabilities(torch.log(probs), frames, torch.log(trans),
                                                            torch.log(init), log_probs=True, num_threads=1).numpy()

    # ---- from_files_to_files (with the configured BATCH_SIZE: several batches; chunked or not)
    lengths = [9, 30, 1, 17, 44, 5, 23] if not chunking else [60, 9, 150, 33]
    ins, outs = [], []
    for k, n in enumerate(lengths):
        certain = [] if not chunking else [t for t in range(n) if (t % 13) in (5, 6) or (k == 2 and t % 29 == 28)]
        x = near_certain(n, S, certain)
        f = os.path.join(work, f'in{k}.pt')
        torch.save(x, f)
        ins.append(f)
        outs.append(os.path.join(work, f'out{k}.pt'))
        out[f'files_{tag}/in{k}'] = x.numpy()
    transition = torch.rand(S, S, generator=gen).softmax(-1)
    tf = os.path.join(work, 'transition.pt')
    torch.save(transition, tf)
    out[f'files_{tag}/transition'] = transition.numpy()
    out[f'files_{tag}/count'] = np.asarray(len(lengths))
    out[f'files_{tag}/batch_size'] = np.asarray(torbi.BATCH_SIZE)
    torbi.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=None, num_threads=1)
    for k, f in enumerate(outs):
        out[f'files_{tag}/out{k}'] = torch.load(f).numpy()

    if chunking:
        mod = sys.modules['torbi.chunk']
        out['chunk/min_chunk_size'] = np.asarray(torbi.MIN_CHUNK_SIZE)
        out['chunk/entropy_threshold'] = np.asarray(torbi.ENTROPY_THRESHOLD)
        for k in range(len(lengths)):
            x = torch.load(ins[k])
            out[f'chunk/entropy{k}'] = mod.entropy(x.T).numpy()
            out[f'chunk/split{k}'] = np.asarray(mod.split(x), dtype=np.int64)
            out[f'chunk/pieces{k}'] = np.asarray([piece.shape[0] for piece in torbi.chunk(x)], dtype=np.int64)
        batch = [torbi.data.Dataset(ins)[k] for k in range(len(lengths))]
        observation, batch_frames, batch_chunks, _ = torbi.data.collate(batch)
        out['chunk/collate_frames'] = batch_frames.numpy()
        out['chunk/collate_chunks'] = np.asarray(batch_chunks)
        out['chunk/collate_shape'] = np.asarray(observation.shape)
        # other thresholds / sizes through the explicit arguments
        x = torch.load(ins[2])
        for size, thr in [(1, 0.5), (5, 0.9), (40, 0.5), (7, 0.05)]:
            out[f'chunk/split2_size{size}_thr{thr}'] = np.asarray(
                mod.split(x, min_chunk_size=size, entropy_threshold=thr), dtype=np.int64)
    np.savez(result_file, **out)


def main():
    where = tempfile.mkdtemp(prefix='torbi_refpkg_')
    try:
        assemble(where)
        merged = {}
        for env in ({'TORBI_STUB_BATCH_SIZE': '3'}, {'TORBI_STUB_BATCH_SIZE': '3', 'TORBI_STUB_MIN_CHUNK': '8'}):
            result = os.path.join(where, 'result_%d.npz' % len(merged))
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', where, result],
                                  env={**os.environ, **env})
            with np.load(result) as data:
                merged.update({k: data[k] for k in data.files})
        np.savez_compressed(OUT, **merged)
        print('wrote', OUT, f'({os.path.getsize(OUT) / 1024:.1f} KiB, {len(merged)} arrays)')
        summary = {k: merged[k].tolist() for k in merged if k.startswith('chunk/split') or k.startswith('chunk/pieces')}
        print(json.dumps(summary))
    finally:
        shutil.rmtree(where, ignore_errors=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(sys.argv[2], sys.argv[3])
    else:
        main()
