"""The host twin of the operator (include/torbi_cpu.h, torbi_amd.decode_cpu) -- what `gpu=None` selects, as in the
reference (torbi/core.py:147-150).  Its own parity tests: the reference operator's golden vectors, the CPU oracle on
fresh inputs, and the reference's Python outputs for the API entry points (all without a GPU).  The twin is not the
oracle and not a fallback: see test_gpu_requests_never_reach_the_twin.
"""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest
import torch

import oracle
import torbi_amd
from torbi_amd import _lib, synth
from conftest import SMALL_NAMES, LARGE_NAMES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
API = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_api.npz'))


def twin(obs, frames, trans, init, num_threads=None):
    return torbi_amd.decode_cpu(torch.as_tensor(np.ascontiguousarray(obs, dtype=np.float32)),
                                torch.as_tensor(np.asarray(frames, dtype=np.int32)),
                                torch.as_tensor(np.ascontiguousarray(trans, dtype=np.float32)),
                                torch.as_tensor(np.ascontiguousarray(init, dtype=np.float32)),
                                num_threads=num_threads).numpy()


def test_library_exports_what_the_header_declares():
    header = open(os.path.join(ROOT, 'include', 'torbi_cpu.h')).read()
    declared = set(re.findall(r'\b(torbi_cpu_[a-z_]+)\s*\(', header))
    assert declared == {'torbi_cpu_abi_version', 'torbi_cpu_viterbi_decode', 'torbi_cpu_read_rows', 'torbi_cpu_write_files', 'torbi_cpu_open_heads'}
    lib = ctypes.CDLL(_lib.CPU_LIBRARY)
    for name in declared:
        getattr(lib, name)
    assert _lib.load_cpu().torbi_cpu_abi_version() == _lib.CPU_ABI_VERSION


@pytest.mark.parametrize('name', SMALL_NAMES)
def test_twin_matches_the_reference_operator_small(golden, name):
    obs, frames, trans, init, want = golden.small_case(name)
    for threads in (1, 3):
        got = twin(obs, frames, trans, init, threads)
        assert got.dtype == np.int32 and got.shape == want.shape
        assert np.array_equal(got, want), threads


@pytest.mark.parametrize('name', LARGE_NAMES)
def test_twin_matches_the_reference_operator_large(golden, name):
    obs, frames, trans, init, want = golden.large_case(name)
    got = twin(obs, frames, trans, init)
    assert np.array_equal(got, want)
    assert hashlib.sha256(got.tobytes()).hexdigest() == str(golden.large[name + '/sha256'])


@pytest.mark.parametrize('seed', range(10))
def test_twin_equals_the_oracle_on_fresh_inputs(seed):
    """Ragged lengths, batches around the item-block size, state counts that are no multiple of the vector width,
    heavy ties, -inf transitions and observations; every thread count, both threading schemes."""
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 2, 7, 8, 9, 17, 24]))
    T, S = int(rng.integers(1, 40)), int(rng.choice([1, 3, 15, 16, 17, 70, 129, 300]))
    obs, trans, init = synth.problem(B, T, S, seed=2000 + seed)
    if seed % 3 == 1:       # heavy ties
        obs, trans, init = np.round(obs / 4), np.round(trans / 4), np.round(init / 4)
    if seed % 3 == 2:       # -inf bands and observation rows
        trans = trans.copy()
        trans[np.abs(np.subtract.outer(np.arange(S), np.arange(S))) > max(1, S // 5)] = -np.inf
        obs = obs.copy()
        obs[0, T // 2, :] = -np.inf
        obs[B - 1, :, S // 2] = -np.inf
    frames = rng.integers(1, T + 1, B).astype(np.int32)
    frames[0] = T
    want = oracle.decode(obs, frames, trans, init, num_threads=2)
    for threads in (1, 2, 5):
        assert np.array_equal(twin(obs, frames, trans, init, threads), want), (B, T, S, threads)


@pytest.mark.parametrize('kind', ['observation', 'matrix', 'initial', 'final_row', 'inf'])
@pytest.mark.parametrize('threads', [1, 5])
def test_twin_follows_the_reference_on_nan_and_inf_inputs(kind, threads):
    """The reference is deterministic on NaN (a NaN candidate at prev-state 0 is never replaced, one elsewhere never wins,
    viterbi.cpp:94-100; the final state is ATen's argmax, the first NaN of the last row, :218), a vectorised maximum is not: the
    twin decodes the items that produced a NaN / +inf posterior value again in the reference's order.  The cases of
    tests/test_oracle.py (where the oracle is pinned against the reference operator itself) on 20 items, one team and several."""
    B, T, S = 20, 12, 40
    obs, trans, init = synth.problem(B, T, S, seed=2100)
    nan = np.float32('nan')
    frames = np.clip(synth.lengths(B, 2, T, seed=1), 2, T).astype(np.int32)
    frames[0] = T
    if kind == 'observation':
        obs[0, 3, 5] = nan
        obs[17, 1, 0] = nan
    elif kind == 'matrix':
        trans[7, 0] = nan
        trans[9, 11] = nan
    elif kind == 'initial':
        init[0] = nan
    elif kind == 'final_row':
        obs[1, frames[1] - 1, 17] = nan
        obs[2, 0, :] = nan
    else:
        obs[0, 2, 0] = np.inf
        obs[1, 1, 3] = np.inf
        trans[6, 0] = -np.inf
        trans[5, 3] = -np.inf
    want = oracle.decode(obs, frames, trans, init, num_threads=2)
    got = torbi_amd.decode_cpu(torch.tensor(obs), torch.tensor(frames), torch.tensor(trans), torch.tensor(init), num_threads=threads)
    np.testing.assert_array_equal(got.numpy(), want)


def test_twin_clamps_lengths_and_fills_the_tail():
    obs, trans, init = synth.problem(3, 10, 7, seed=9)
    got = twin(obs, [4, 0, 99], trans, init)
    assert (got[0, 3:] == got[0, 3]).all()
    assert np.array_equal(got[1], twin(obs[1:2], [1], trans, init)[0])       # 0 -> 1 like the device path
    assert np.array_equal(got[2], twin(obs[2:3], [10], trans, init)[0])      # > T -> T
    assert torbi_amd.decode_cpu(torch.zeros(0, 5, 3), torch.zeros(0, dtype=torch.int32), torch.zeros(3, 3),
                                torch.zeros(3)).shape == (0, 5)


def test_twin_validates_like_the_operator():
    with pytest.raises(RuntimeError, match='expected scalar type'):
        torbi_amd.decode_cpu(torch.zeros(1, 3, 3), torch.tensor([3]), torch.zeros(3, 3), torch.zeros(3))
    with pytest.raises(RuntimeError, match='transition must have shape'):
        torbi_amd.decode_cpu(torch.zeros(1, 3, 3), torch.tensor([3], dtype=torch.int32), torch.zeros(3, 4), torch.zeros(3))


def test_gpu_none_runs_the_reference_toys_on_the_cpu():
    """reference tests/test_core.py:7-46 and the defaults SURVEY 8c lists: gpu=None is the CPU route."""
    observation = torch.tensor([[0.25, 0.5, 0.25], [0.25, 0.25, 0.5], [0.33, 0.33, 0.33]]).unsqueeze(dim=0)
    transition = torch.tensor([[0.5, 0.25, 0.25], [0.33, 0.34, 0.33], [0.25, 0.25, 0.5]])
    initial = torch.tensor([0.4, 0.35, 0.25])
    bins = torbi_amd.from_probabilities(observation=observation, transition=transition, initial=initial, log_probs=False)
    assert bins.device.type == 'cpu' and bins.dtype == torch.int32 and bins.tolist() == [[1, 2, 2]]
    assert torbi_amd.from_probabilities(observation).tolist() == [[1, 2, 0]]
    assert torbi_amd.from_probabilities(observation, batch_frames=torch.tensor([2]), transition=transition,
                                        initial=initial).tolist() == [[1, 2, 2]]


def test_from_probabilities_on_the_cpu_equals_the_reference_outputs():
    """SURVEY 8c G7 on the reference's own device: its from_probabilities (real torbi Python, CPU operator) on
    probability and log-probability inputs, given and default models (tests/golden/generate_api.py)."""
    obs = torch.as_tensor(API['probs/observation'])
    trans = torch.as_tensor(API['probs/transition'])
    init = torch.as_tensor(API['probs/initial'])
    frames = torch.as_tensor(API['probs/batch_frames'])
    got = torbi_amd.from_probabilities(obs.clone(), frames, trans, init, log_probs=False)
    np.testing.assert_array_equal(got.numpy(), API['probs/indices'])
    np.testing.assert_array_equal(torbi_amd.from_probabilities(obs.clone()).numpy(), API['probs/indices_defaults'])
    got = torbi_amd.from_probabilities(torch.log(obs), frames, torch.log(trans), torch.log(init), log_probs=True,
                                       num_threads=2)
    np.testing.assert_array_equal(got.numpy(), API['probs/indices_log'])


@pytest.mark.parametrize('tag', ['plain', 'chunk'])
def test_from_files_to_files_on_the_cpu_equals_the_reference_outputs(tmp_path, monkeypatch, tag):
    """SURVEY 8c G6 / 8f rank 4 without a GPU: the files the reference's from_files_to_files wrote (plain, and with
    chunked decoding at MIN_CHUNK_SIZE = 8) -- same shapes, dtypes and indices from gpu=None here."""
    monkeypatch.setattr(torbi_amd.core, 'BATCH_SIZE', int(API[f'files_{tag}/batch_size']))
    if tag == 'chunk':
        monkeypatch.setattr(torbi_amd.core, 'MIN_CHUNK_SIZE', int(API['chunk/min_chunk_size']))
    count = int(API[f'files_{tag}/count'])
    ins, outs = [], []
    for k in range(count):
        f = tmp_path / f'in{k}.pt'
        torch.save(torch.as_tensor(API[f'files_{tag}/in{k}']), f)
        ins.append(f)
        outs.append(tmp_path / f'out{k}.pt')
    tf = tmp_path / 'transition.pt'
    torch.save(torch.as_tensor(API[f'files_{tag}/transition']), tf)
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, num_threads=2)
    for k, f in enumerate(outs):
        got = torch.load(f)
        want = API[f'files_{tag}/out{k}']
        assert got.dtype == torch.int32 and tuple(got.shape) == want.shape
        np.testing.assert_array_equal(got.numpy(), want, err_msg=f'file {k}')


def test_gpu_requests_never_reach_the_twin(monkeypatch):
    """No fallback: with the twin made unusable, gpu=None fails and nothing else notices; with no HIP device a GPU
    request raises instead of decoding on the CPU."""
    obs = torch.full((1, 3, 3), 1 / 3)

    def broken():
        raise AssertionError('the CPU twin was reached')

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no CPU\\s+fallback'):
            torbi_amd.from_probabilities(obs, gpu=0)
        with pytest.raises(RuntimeError, match='HIP device'):
            torbi_amd.decode(torch.zeros(1, 3, 3), torch.tensor([3], dtype=torch.int32), torch.zeros(3, 3), torch.zeros(3))
    monkeypatch.setattr(_lib, 'load_cpu', broken)
    with pytest.raises(AssertionError, match='twin was reached'):
        torbi_amd.from_probabilities(obs)
    if torch.cuda.is_available():
        assert torbi_amd.from_probabilities(obs, gpu=0).tolist() == torbi_amd.from_probabilities(obs.cuda(), gpu=0).tolist()
    sources = [f for f in os.listdir(os.path.join(ROOT, 'torbi_amd', 'csrc')) if f != 'torbi_cpu.cpp']
    for name in sources:
        assert 'torbi_cpu' not in open(os.path.join(ROOT, 'torbi_amd', 'csrc', name)).read(), name


def test_command_line_without_gpu_decodes_on_the_cpu_like_upstream(tmp_path):
    """`python -m torbi_amd --input_files ... --output_files ... --transition_file ... --log_probs` (reference
    torbi/__main__.py:16-49; no --gpu = the CPU operator): the reference's own output files."""
    import subprocess
    import sys
    count = int(API['files_plain/count'])
    ins, outs = [], []
    for k in range(count):
        f = tmp_path / f'in{k}.pt'
        torch.save(torch.as_tensor(API[f'files_plain/in{k}']), f)
        ins.append(str(f))
        outs.append(str(tmp_path / f'out{k}.pt'))
    tf = tmp_path / 'transition.pt'
    torch.save(torch.as_tensor(API['files_plain/transition']), tf)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''), CUDA_VISIBLE_DEVICES='',
               HIP_VISIBLE_DEVICES='')
    done = subprocess.run([sys.executable, '-m', 'torbi_amd', '--input_files', *ins, '--output_files', *outs,
                           '--transition_file', str(tf), '--log_probs', '--num_threads', '2'],
                          cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    for k, f in enumerate(outs):
        got = torch.load(f)
        np.testing.assert_array_equal(got.numpy(), API[f'files_plain/out{k}'], err_msg=f'file {k}')


def test_dispatcher_registration_serves_cpu_tensors():
    """reference torbi/viterbi.py:53 on CPU tensors: torch.ops.torbi.viterbi_decode reaches the host twin (the CPU key
    the reference registers at torbi/csrc/viterbi.cpp:237-239); int64 lengths are rejected like upstream.  In a process
    of its own: the registration claims the `torbi` operator namespace, which the reference's own library (loaded by the
    oracle tests as a checker) claims too."""
    import subprocess
    import sys
    script = """
import numpy as np, torch
import oracle
from torbi_amd import synth, torch_op
op = torch_op.register()
obs, trans, init = synth.problem(3, 9, 20, seed=8)
frames = np.array([9, 4, 1], dtype=np.int32)
got = op(torch.as_tensor(obs), torch.as_tensor(frames), torch.as_tensor(trans), torch.as_tensor(init))
assert got.dtype == torch.int32 and np.array_equal(got.numpy(), oracle.decode(obs, frames, trans, init))
try:
    op(torch.as_tensor(obs), torch.as_tensor(frames.astype(np.int64)), torch.as_tensor(trans), torch.as_tensor(init))
except RuntimeError as exc:
    assert 'expected scalar type' in str(exc)
else:
    raise AssertionError('int64 lengths were accepted')
print('dispatcher ok')
"""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    done = subprocess.run([sys.executable, '-c', script], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and 'dispatcher ok' in done.stdout, done.stderr[-2000:]
