"""Host-side logic and the C-ABI surface, without a GPU (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import torbi_amd
from torbi_amd import _lib, synth, distributed, viterbi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'torbi_hip.h')).read()
    declared = set(re.findall(r'\b(torbi_hip_\w+)\s*\(', header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.torbi_hip_abi_version() == _lib.ABI_VERSION


def test_every_named_forward_path_is_accepted_by_the_library():
    """The header's path constants, the Python names and the setter's range agree (no GPU needed: the setter only
    stores the process default); the path flag of a call has room for every path."""
    import ctypes
    from torbi_amd import viterbi
    header = open(os.path.join(ROOT, 'include', 'torbi_hip.h')).read()
    constants = {name.lower(): int(value) for name, value in re.findall(r'#define TORBI_HIP_FORWARD_(\w+) (\d+)', header)}
    assert constants == viterbi.FORWARD_PATHS
    lib = _lib.load()
    try:
        for name, code in viterbi.FORWARD_PATHS.items():
            assert lib.torbi_hip_set_forward_path(code) == 0, name
            assert (code + 1) << 4 <= 7 << 4
        assert lib.torbi_hip_set_forward_path(max(constants.values()) + 1) == -1
        assert lib.torbi_hip_set_forward_path(-1) == -1
    finally:
        lib.torbi_hip_set_forward_path(0)
    # every named path is a route of its own, except 'pruned': the recurrence all the time-resident forms and the rows route
    # run (its per-timestep tile kernel, route 2, was removed in round 4)
    assert set(viterbi.ROUTES.values()) >= set(viterbi.FORWARD_PATHS) - {'auto', 'pruned'} and 2 not in viterbi.ROUTES


def test_workspace_bytes_and_error_strings():
    lib = _lib.load()
    need = lib.torbi_hip_workspace_bytes(512, 500, 1440)
    assert need >= 512 * 500 * 1440 * 4 + 2 * 512 * 1440 * 4
    assert need % 256 == 0
    assert lib.torbi_hip_workspace_bytes(0, 0, 0) > 0
    assert b'success' in lib.torbi_hip_error_string(0)
    assert b'workspace' in lib.torbi_hip_error_string(-2)


def test_argument_errors_without_touching_a_device():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.torbi_hip_viterbi_decode(null, null, null, null, null, null, 0, 1, 1, 1, 0, null) == -1
    assert lib.torbi_hip_viterbi_decode(null, null, null, null, null, null, 0, 1, 0, 1, 0, null) == -1
    buf = (ctypes.c_char * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.torbi_hip_viterbi_decode(p, p, p, p, p, p, 64, 4, 4, 4, 0, null) == -2
    assert lib.torbi_hip_viterbi_decode(null, null, null, null, null, null, 0, 0, 1, 1, 0, null) == 0


def test_decode_validates_like_the_reference_operator():
    obs = torch.zeros(1, 3, 3)
    trans = torch.zeros(3, 3)
    init = torch.zeros(3)
    with pytest.raises(RuntimeError, match='batch_frames'):   # int64 lengths are rejected upstream too
        torbi_amd.decode(obs, torch.tensor([3]), trans, init)
    with pytest.raises(RuntimeError, match='observation'):
        torbi_amd.decode(obs.double(), torch.tensor([3], dtype=torch.int32), trans, init)
    with pytest.raises(RuntimeError, match='shape'):
        torbi_amd.decode(obs[0], torch.tensor([3], dtype=torch.int32), trans, init)
    with pytest.raises(RuntimeError, match='transition'):
        torbi_amd.decode(obs, torch.tensor([3], dtype=torch.int32), torch.zeros(3, 4), init)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_no_gpu_fails_loudly_instead_of_falling_back():
    """A GPU request without a HIP device raises; it is never served by the CPU operator (which only `gpu=None`
    selects, like upstream -- tests/test_cpu_twin.py)."""
    obs = torch.full((1, 3, 3), 1 / 3)
    with pytest.raises(RuntimeError, match='no CPU'):
        torbi_amd.from_probabilities(obs, gpu=0)
    with pytest.raises(RuntimeError, match='HIP device'):
        torbi_amd.decode(torch.zeros(1, 3, 3), torch.tensor([3], dtype=torch.int32),
                         torch.zeros(3, 3), torch.zeros(3))
    with pytest.raises(RuntimeError, match='HIP device'):
        torbi_amd.DecodePipeline()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'torbi_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', text, re.M), f
                assert 'libviterbi_oracle' not in text, f


def test_synth_is_deterministic_and_exact():
    a = synth.scores(1, (3, 5), seed=0)
    b = synth.scores(1, (15,), seed=0).reshape(3, 5)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert (a <= 0).all() and (a > -16).all()
    # values are 24-bit integers times 2**-20
    assert np.array_equal(a * 2 ** 20, np.round(a * 2 ** 20))
    # windows of one stream agree
    assert np.array_equal(synth.scores(2, (10,), start=5), synth.scores(2, (15,))[5:])
    lens = synth.lengths(1000, 100, 900)
    assert lens.min() >= 100 and lens.max() <= 900


def test_synth_golden_values():
    """Frozen hash outputs: guards the generator the committed goldens depend on."""
    assert synth.hash_u24(1, 0, 4).tolist() == [14819496, 9505325, 9918517, 1903380]
    assert synth.hash_u24(2, 7, 3, seed=5).tolist() == [6252287, 11286206, 8102033]
    assert synth.scores(3, (3,)).tolist() == [
        -0.4229402542114258, -15.536043167114258, -9.530208587646484]


def test_collate_pads_and_separate_rejoins(tmp_path):
    """reference torbi/data/collate.py:9-45"""
    items = [(torch.arange(6, dtype=torch.float32).reshape(3, 2), 'a'),
             (torch.ones(5, 2), 'b'), (torch.full((1, 2), 7.), 'c')]
    obs, frames, chunks, files = torbi_amd.data.collate(items)
    assert obs.shape == (3, 5, 2) and frames.tolist() == [3, 5, 1]
    assert chunks == [1, 1, 1] and files == ('a', 'b', 'c')
    assert torch.equal(obs[0, :3], items[0][0]) and (obs[0, 3:] == 0).all() and (obs[2, 1:] == 0).all()
    with pytest.raises(ValueError):
        torbi_amd.data.collate([])
    idx = torch.arange(15).reshape(3, 5)
    parts = torbi_amd.data.separate(idx, [2, 1], torch.tensor([3, 5, 1]))
    assert parts[0].tolist() == [0, 1, 2, 5, 6, 7, 8, 9] and parts[1].tolist() == [10]


def test_loader_batches_in_order(tmp_path):
    files = []
    for k in range(5):
        f = tmp_path / f'{k}.pt'
        torch.save(torch.full((k + 1, 3), float(k)), f)
        files.append(f)
    batches = list(torbi_amd.data.loader(files, batch_size=2))
    assert [len(b[3]) for b in batches] == [2, 2, 1]
    assert batches[1][1].tolist() == [3, 4] and batches[1][0].shape == (2, 4, 3)


def test_save_masked(tmp_path):
    """reference torbi/core.py:471-473"""
    f = tmp_path / 'o.pt'
    torbi_amd.save_masked(torch.arange(10, dtype=torch.int32), f, torch.tensor(4))
    assert torch.load(f).tolist() == [0, 1, 2, 3]


def test_shard_bounds_cover_everything():
    for count in (0, 1, 7, 512, 513):
        for size in (1, 2, 3, 8):
            spans = [distributed.shard_bounds(count, size, r) for r in range(size)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_assign_batches_is_a_partition():
    lengths = synth.lengths(2000, 100, 900).tolist()
    plan = distributed.assign_batches(lengths, 512, 3)
    flat = sorted(i for rank in plan for batch in rank for i in batch)
    assert flat == list(range(2000))
    for rank in plan:
        for batch in rank:
            assert batch == list(range(batch[0], batch[0] + len(batch))) and len(batch) <= 512


def test_auto_path_choice_reads_the_transition_structure_once_per_version():
    """torbi_amd/viterbi.py::_choose_path: narrow bands -> dense kernel (-inf block skipping), everything else ->
    pruned; forced paths and unsupported shapes pass through; the look is cached per tensor version."""
    import torch
    from torbi_amd import state, synth, viterbi
    S = 320         # (up to 256 states AUTO is a workgroup per sequence whatever the matrix looks like: csrc/small_states.hpp)
    dense = torch.as_tensor(synth.problem(1, 1, S, seed=1)[1])
    band = torch.as_tensor(synth.banded_transition(S, 12.0))
    assert viterbi._resolve_path(dense, dense, 64, 256, 'cuda:0', 'auto', 4) == 'auto'
    assert viterbi._choose_path(dense, dense, 64, S) == 'pruned'
    assert viterbi._choose_path(band, band, 64, S) == 'dense'
    assert 0.0 < state.notes(band)[('reach', S)] < viterbi.BANDED_RANGE
    band.fill_(-1.0)                                  # new version of the same storage: looked at again
    assert viterbi._choose_path(band, band, 64, S) == 'pruned'
    dead = torch.full((S, S), float('-inf'))
    assert viterbi._choose_path(dead, dead, 64, S) == 'pruned'          # reach 0: nothing to skip *to*
    assert viterbi._choose_path(dense, dense, 8, S) == 'auto'           # small batch: generic kernels either way
    assert viterbi._choose_path(dense, dense, 64, 8192) == 'auto'       # outside the pruned path's range
    # a forced path never reaches the look; an explicit per-call 'auto' ignores the process default
    old, units = viterbi._forced_path, dict(viterbi._compute_units)
    band2 = torch.as_tensor(synth.banded_transition(S, 12.0))     # `band` was overwritten above
    try:
        viterbi._forced_path = 'dense'
        viterbi._compute_units[0] = 256               # what torbi_hip_compute_units(0) reports on an MI355X
        assert viterbi._resolve_path(dense, dense, 64, S, 'cuda:0', None, 4) == 'dense'
        assert viterbi._resolve_path(dense, dense, 64, S, 'cuda:0', 'auto', 4) == 'cluster'
        assert viterbi._resolve_path(dense, dense, 64, S, 'cuda:0', 'resident', 4) == 'resident'
        # enough 16-item tiles to give half the compute units a workgroup: AUTO decodes time-resident ('cluster' = the
        # library picks the form: whole tiles per workgroup here, clusters of workgroups per tile below half the chip)
        assert viterbi._resolve_path(dense, dense, 2048, S, 'cuda:0', 'auto', 128) == 'cluster'
        assert viterbi._resolve_path(band, band, 2048, S, 'cuda:0', 'auto', 128) == 'cluster'
        # below half the chip -- a launch group or one batch of more than 16 items -- clusters of workgroups per tile;
        # one batch with a narrow band: the dense kernel's -inf skipping; 16 items or fewer: the library's choice
        assert viterbi._resolve_path(dense, dense, 512, S, 'cuda:0', 'auto', 64, count=2) == 'cluster'
        assert viterbi._resolve_path(dense, dense, 768, S, 'cuda:0', 'auto', 48) == 'cluster'
        assert viterbi._resolve_path(dense, dense, 512, S, 'cuda:0', 'auto', 32) == 'cluster'
        assert viterbi._resolve_path(band2, band2, 512, S, 'cuda:0', 'auto', 32) == 'dense'
        wide = torch.as_tensor(synth.banded_transition(2064, 12.0))
        assert viterbi._resolve_path(wide, wide, 64, 2064, 'cuda:0', 'auto', 8) == 'dense'    # 8-item tiles
        assert viterbi._resolve_path(band2, band2, 512, S, 'cuda:0', 'auto', 32, count=2) == 'cluster'
        assert viterbi._resolve_path(dense, dense, 16, S, 'cuda:0', 'auto', 1) == 'auto'
    finally:
        viterbi._forced_path = old
        viterbi._compute_units.clear()
        viterbi._compute_units.update(units)
    with torch.inference_mode():                      # no version counter: looked at, never cached
        inf = torch.as_tensor(synth.banded_transition(S, 12.0)) * 1
        assert viterbi._version_of(inf) is None
        assert viterbi._choose_path(inf, inf, 64, S) == 'dense'
        assert state.peek(inf) is None and state.notes(inf) is None


# ---- vectors produced by the reference's own Python (tests/golden/generate_api.py) ---------------------------

API = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_api.npz'))


def test_collate_equals_the_reference_collate():
    """reference torbi/data/collate.py:9-33 on ragged items: same padded tensor, lengths, chunk counts, names."""
    items = [(torch.as_tensor(API[f'collate/item{k}']), str(API['collate/names'][k])) for k in range(4)]
    observation, batch_frames, batch_chunks, names = torbi_amd.data.collate(items)
    assert np.array_equal(observation.numpy(), API['collate/observation'])
    assert observation.dtype == torch.float32
    assert batch_frames.tolist() == API['collate/batch_frames'].tolist()
    assert batch_frames.dtype == torch.int64            # upstream builds it with torch.tensor(list of ints)
    assert list(batch_chunks) == API['collate/batch_chunks'].tolist()
    assert list(names) == [str(n) for n in API['collate/names']]


def test_chunk_cut_points_equal_the_reference(monkeypatch):
    """reference torbi/chunk.py:12-85 (real Python, MIN_CHUNK_SIZE = 8): entropy, cut points and piece lengths
    of four files; other sizes / thresholds through the explicit arguments."""
    chunk_module = __import__('torbi_amd.chunk', fromlist=['split'])
    size, thr = int(API['chunk/min_chunk_size']), float(API['chunk/entropy_threshold'])
    assert (size, thr) == (8, torbi_amd.core.ENTROPY_THRESHOLD)
    monkeypatch.setattr(torbi_amd.core, 'MIN_CHUNK_SIZE', size)
    for k in range(4):
        x = torch.as_tensor(API[f'files_chunk/in{k}'])
        # entropy values to the last bits of the host's exp / log / sum kernels (they differ between an AVX2 and an
        # AVX-512 build of torch's CPU ops); the cut points below are the contract and stay exact
        np.testing.assert_allclose(chunk_module.entropy(x).numpy(), API[f'chunk/entropy{k}'], rtol=2e-5, atol=1e-12)
        assert chunk_module.split(x, size, thr) == API[f'chunk/split{k}'].tolist()
        pieces = torbi_amd.chunk(x)                     # defaults read from core at call time
        assert [p.shape[0] for p in pieces] == API[f'chunk/pieces{k}'].tolist()
        assert torch.equal(torch.cat(pieces), x)
    x = torch.as_tensor(API['files_chunk/in2'])
    for size, thr in [(1, 0.5), (5, 0.9), (40, 0.5), (7, 0.05)]:
        assert chunk_module.split(x, size, thr) == API[f'chunk/split2_size{size}_thr{thr}'].tolist()
    with pytest.raises(ValueError):
        chunk_module.split(x, None, 0.5)
    assert chunk_module.split(torch.full((30, 4), float('-inf')), 2, 0.5) == []      # NaN entropy never qualifies


def test_chunked_dataset_and_collate_equal_the_reference(tmp_path, monkeypatch):
    """dataset.py:22-23 + collate.py:13-15: files cut into pieces become consecutive batch rows."""
    monkeypatch.setattr(torbi_amd.core, 'MIN_CHUNK_SIZE', int(API['chunk/min_chunk_size']))
    files = []
    for k in range(4):
        f = tmp_path / f'in{k}.pt'
        torch.save(torch.as_tensor(API[f'files_chunk/in{k}']), f)
        files.append(f)
    dataset = torbi_amd.data.Dataset(files)
    observation, batch_frames, batch_chunks, names = torbi_amd.data.collate([dataset[k] for k in range(4)])
    assert batch_frames.tolist() == API['chunk/collate_frames'].tolist()
    assert list(batch_chunks) == API['chunk/collate_chunks'].tolist()
    assert list(observation.shape) == API['chunk/collate_shape'].tolist()
    assert list(names) == files


def test_from_dataloader_joins_chunked_files(tmp_path, monkeypatch):
    """reference torbi/core.py:438-448: with chunked items every FILE gets the concatenation of its rows' valid
    frames (a stand-in decode labels every position with 1000 * row + frame, no GPU involved)."""
    monkeypatch.setattr(torbi_amd.core, 'MIN_CHUNK_SIZE', 8)
    monkeypatch.setattr(torbi_amd.core, 'BATCH_SIZE', 3)
    files, mapping, lengths = [], {}, {}
    for k in range(4):
        f = tmp_path / f'in{k}.pt'
        x = torch.as_tensor(API[f'files_chunk/in{k}'])
        torch.save(x, f)
        files.append(f)
        mapping[f] = tmp_path / f'out{k}.pt'
        lengths[f] = API[f'chunk/pieces{k}'].tolist()

    def fake_from_probabilities(observation, batch_frames, **_):
        rows, frames = observation.shape[:2]
        return (1000 * torch.arange(rows)[:, None] + torch.arange(frames)[None, :]).to(torch.int32)

    monkeypatch.setattr(torbi_amd.core, 'from_probabilities', fake_from_probabilities)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    torbi_amd.from_dataloader(torbi_amd.data.loader(files, pin_memory=False), mapping)
    row = 0
    for k, f in enumerate(files):
        if k == 3:
            row = 0                                     # second batch of the loader (batch size 3 files)
        want = []
        for n in lengths[f]:
            want.extend(1000 * row + t for t in range(n))
            row += 1
        got = torch.load(mapping[f])
        assert got.dtype == torch.int32 and got.tolist() == want


# ---- direct file reader (torbi_amd/fastio.py): same batches as torch.load + collate ------------------------------

def _write_ragged_files(folder, lengths, states=12, seed=0):
    generator = torch.Generator().manual_seed(seed)
    files = []
    for k, n in enumerate(lengths):
        f = str(folder / f'in{k}.pt')
        torch.save(torch.rand(n, states, generator=generator), f)
        files.append(f)
    return files


def _same_batches(ours, theirs):
    ours, theirs = list(ours), list(theirs)
    assert len(ours) == len(theirs)
    for a, b in zip(ours, theirs):
        assert a[0].dtype == b[0].dtype and torch.equal(a[0], b[0])
        assert a[1].dtype == b[1].dtype and torch.equal(a[1], b[1])
        assert list(a[2]) == list(b[2]) and tuple(a[3]) == tuple(b[3])


def test_file_batches_equal_the_loader_batches(tmp_path):
    """The payload of every file lands in its row, the rest of the row is collate's zero padding
    (reference collate.py:24-31), lengths / chunk counts / names as the DataLoader yields them."""
    from torbi_amd import fastio
    files = _write_ragged_files(tmp_path, [5, 1, 9, 3, 7, 2, 8, 4])
    assert fastio.payload(files[2])[:2] == (9, 12)
    for batch_size in (1, 3, 8, 20):
        _same_batches(fastio.FileBatches(files, batch_size, threads=3, pin_memory=False),
                      torbi_amd.data.loader(files, num_workers=0, batch_size=batch_size, pin_memory=False))
    batches = fastio.open_batches(files, 3, threads=2)
    assert isinstance(batches, fastio.FileBatches) and len(batches) == 3


def test_payload_found_from_the_head_of_the_file_equals_the_central_directory_route(tmp_path):
    """fastio.payload_of_open_file walks the local headers in the first 4 KB (torch leaves record sizes to data
    descriptors); fastio.payload asks the zip central directory.  Same answer, also for a view with a storage
    offset; a file cut short is reported by the native reader, not decoded."""
    from torbi_amd import fastio
    f = str(tmp_path / 'x.pt')
    base = torch.rand(40, 12, generator=torch.Generator().manual_seed(3))
    for tensor in (torch.rand(500, 1440), torch.rand(1, 3), torch.rand(33, 4096), base[4:9], base[39:]):
        torch.save(tensor, f)
        fd = os.open(f, os.O_RDONLY)
        try:
            found = fastio.payload_of_open_file(fd)
        finally:
            os.close(fd)
        assert found == fastio.payload(f) and found[:2] == tuple(tensor.shape)
        raw = np.fromfile(f, dtype=np.uint8)[found[2]:found[2] + 4 * tensor.numel()].view(np.float32)
        assert np.array_equal(raw.reshape(tensor.shape), tensor.numpy())
    torch.save(torch.rand(50, 12), f)
    frames, states, start = fastio.payload(f)
    with open(f, 'r+b') as handle:
        handle.truncate(start + 4 * 12 * 20)
    # a file cut short: the head route refuses it (the tensor does not fit the file), the batch goes the reference's way
    # and torch.load raises as it would upstream; the native reader reports a short read should a file shrink later
    with pytest.raises(RuntimeError):
        list(fastio.FileBatches([f], 4, pin_memory=False))
    import ctypes
    read_rows, _ = torbi_amd._lib.host_io(False)
    fd = os.open(f, os.O_RDONLY)
    try:
        row = np.zeros(50 * 12, np.float32)
        args = [np.array([fd], np.int32), np.array([start], np.int64), np.array([4 * 12 * 50], np.int64),
                np.array([row.ctypes.data], np.int64), np.array([0], np.int64)]
        error = ctypes.c_int(0)
        assert read_rows(*[a.ctypes.data for a in args], 1, 1, ctypes.byref(error)) == -100 and error.value == 0
    finally:
        os.close(fd)


def test_file_batches_equal_the_reference_collate(tmp_path):
    """The reference's own collate output (tests/golden/generate_api.py) from files holding its inputs."""
    from torbi_amd import fastio
    files = []
    for k in range(4):
        f = str(tmp_path / f'{k}.pt')
        torch.save(torch.as_tensor(API[f'collate/item{k}']), f)
        files.append(f)
    (observation, batch_frames, batch_chunks, names), = list(fastio.FileBatches(files, 4, pin_memory=False))
    assert observation.dtype == torch.float32 and np.array_equal(observation.numpy(), API['collate/observation'])
    assert batch_frames.dtype == torch.int64 and batch_frames.tolist() == API['collate/batch_frames'].tolist()
    assert list(batch_chunks) == API['collate/batch_chunks'].tolist() and list(names) == files


def test_file_batches_leave_unusual_files_to_torch_load(tmp_path, monkeypatch):
    """Views with a storage offset are read in place; other dtypes, non-contiguous tensors and the legacy
    container make their batch go through torch.load + collate; chunked decoding keeps the reference's loader."""
    from torbi_amd import fastio
    files = _write_ragged_files(tmp_path, [6, 4, 7, 3, 5, 2])
    base = torch.rand(20, 12, generator=torch.Generator().manual_seed(5))
    torch.save(base[4:9], files[1])                                   # storage offset 48, 5 frames
    assert fastio.payload(files[1])[:2] == (5, 12)
    _same_batches(fastio.FileBatches(files, 3, threads=2, pin_memory=False),
                  torbi_amd.data.loader(files, num_workers=0, batch_size=3, pin_memory=False))
    torch.save(base[:3].double(), files[3])
    torch.save(base[:12, :12].t(), files[4])                          # stride (1, 12)
    torch.save(base[:2].clone(), files[5], _use_new_zipfile_serialization=False)
    for odd in files[3:]:
        with pytest.raises(fastio.UnsupportedFile):
            fastio.payload(odd)
    _same_batches(fastio.FileBatches(files, 3, threads=2, pin_memory=False),
                  torbi_amd.data.loader(files, num_workers=0, batch_size=3, pin_memory=False))
    assert fastio.open_batches(files, 3) is None                      # last file: legacy container
    assert fastio.open_batches(files[:3], 3) is not None
    monkeypatch.setattr(torbi_amd.core, 'MIN_CHUNK_SIZE', 8)
    assert fastio.open_batches(files[:3], 3) is None
    assert fastio.open_batches([], 3) is None


def test_from_dataloader_saves_every_file_from_direct_batches(tmp_path, monkeypatch):
    """from_dataloader over FileBatches with saver threads: every output holds its own row's valid frames
    (core.py:449-457); a stand-in decode labels positions, no GPU involved."""
    from torbi_amd import fastio
    lengths = [5, 1, 9, 3, 7, 2, 8]
    files = _write_ragged_files(tmp_path, lengths)
    mapping = {f: str(tmp_path / f'out{k}.pt') for k, f in enumerate(files)}

    def fake_from_probabilities(observation, batch_frames, **_):
        rows, frames = observation.shape[:2]
        return (1000 * torch.arange(rows)[:, None] + torch.arange(frames)[None, :]).to(torch.int32)

    monkeypatch.setattr(torbi_amd.core, 'from_probabilities', fake_from_probabilities)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    for threads in (0, 3):
        monkeypatch.setattr(torbi_amd.core, 'SAVE_THREADS', threads)
        monkeypatch.setattr(torbi_amd.core, 'DIRECT_FILE_IO', threads == 3)     # save_indices / save_masked
        torbi_amd.from_dataloader(fastio.FileBatches(files, 3, threads=2, pin_memory=False), mapping)
        for k, f in enumerate(files):
            got = torch.load(mapping[f])
            assert got.dtype == torch.int32 and got.tolist() == [1000 * (k % 3) + t for t in range(lengths[k])]
            os.remove(mapping[f])


def test_save_indices_writes_what_torch_load_expects(tmp_path):
    """fastio.save_indices leaves a torch.save container (reference output format, torbi/core.py:466-473): torch.load
    returns the int32 vector in every mode, the zip checksums hold, other inputs go through torch.save."""
    import zipfile
    from torbi_amd import fastio
    generator = torch.Generator().manual_seed(2)
    for n in (1, 2, 17, 500, 900, 500):
        x = torch.randint(0, 1440, (n,), dtype=torch.int32, generator=generator)
        f = tmp_path / f'o{n}.pt'
        fastio.save_indices(x, f)
        for kwargs in ({}, {'weights_only': True}, {'mmap': True}):
            got = torch.load(f, **kwargs)
            assert got.dtype == torch.int32 and got.shape == (n,) and torch.equal(got, x)
        assert zipfile.ZipFile(f).testzip() is None
    rows = torch.arange(40, dtype=torch.int32).reshape(4, 10)
    fastio.save_indices(rows[2][:7], tmp_path / 'view.pt')                 # a view into a batch of rows
    assert torch.load(tmp_path / 'view.pt').tolist() == list(range(20, 27))
    for odd in (torch.zeros((0,), dtype=torch.int32), torch.arange(6, dtype=torch.int64), rows):
        fastio.save_indices(odd, tmp_path / 'odd.pt')
        assert torch.equal(torch.load(tmp_path / 'odd.pt'), odd)


def test_save_index_rows_writes_a_batch_of_outputs(tmp_path):
    """fastio.save_index_rows (native writer threads, torbi_hip_write_files): every file holds its row's first
    `length` indices (core.py:449-457); rows the image route does not take fall back to torch.save; a path that cannot
    be created raises."""
    from torbi_amd import fastio
    rows = torch.randint(0, 1440, (40, 90), dtype=torch.int32, generator=torch.Generator().manual_seed(4))
    lengths = [1 + (7 * k) % 90 for k in range(40)]
    files = [tmp_path / f'o{k}.pt' for k in range(40)]
    fastio.save_index_rows(rows, files, lengths, threads=3)
    for k in range(40):
        got = torch.load(files[k])
        assert got.dtype == torch.int32 and torch.equal(got, rows[k, :lengths[k]])
    wide = rows.to(torch.int64)
    fastio.save_index_rows(wide[:3], files[:3], [5, None, 0])
    assert torch.equal(torch.load(files[0]), wide[0, :5]) and torch.equal(torch.load(files[1]), wide[1])
    assert torch.load(files[2]).numel() == 0
    with pytest.raises(OSError):
        fastio.save_index_rows(rows[:1], [tmp_path / 'missing' / 'x.pt'], [4])


def test_bench_starts_its_own_ranks_and_dry_runs_without_a_gpu():
    """`python bench.py --gpus 2` without a launcher: bench.py starts the two ranks itself (child process, gloo
    rendezvous on 127.0.0.1) and, with no HIP device, prints a dry-run line carrying the launch plan."""
    if torch.cuda.device_count() > 0:
        pytest.skip('a HIP device is present: the real benchmark would run')
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5'],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['dry_run'] is True and line['value'] is None
    assert line['steps'] == 20 and line['warmup'] == 5 and line['config']['launch_groups'] == [8, 8, 4]
    # the N > 1 line verifies itself: ranks as the collective saw them, per-rank rates, the gather timed on its own
    report = line['multi_gpu']
    assert report['ranks_seen'] == {'world_size': 2, 'all_reduce_of_ones': 2, 'backend': 'gloo'}
    assert report['per_rank_value'] == [None, None] and report['n1_reference_value'] is None
    assert report['gather_ms_per_batch'] > 0 and report['gather_bytes_per_batch'] == 2 * 512 * 500 * 4
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'c4', '--files', '3000'],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong'
    assert line['decode_only']['batches_per_rank'] == [3, 3]
    assert line['decode_only']['valid_timesteps'] == float(synth.lengths(3000, 100, 900).sum())


def test_command_line_takes_the_reference_flags(monkeypatch, tmp_path):
    """reference torbi/__main__.py:16-49: same flag names and types; the call goes to from_files_to_files."""
    from torbi_amd import __main__ as cli
    seen = {}
    monkeypatch.setattr(torbi_amd, 'from_files_to_files', lambda **kwargs: seen.update(kwargs))
    assert cli.main(['--input_files', 'a.pt', 'b.pt', '--output_files', 'x.pt', 'y.pt', '--transition_file', 't.pt',
                     '--initial_file', 'i.pt', '--log_probs', '--gpu', '3', '--num_threads', '4', '--config', 'ignored.py']) == 0
    assert [str(p) for p in seen['input_files']] == ['a.pt', 'b.pt'] and [str(p) for p in seen['output_files']] == ['x.pt', 'y.pt']
    assert str(seen['transition_file']) == 't.pt' and str(seen['initial_file']) == 'i.pt'
    assert seen['log_probs'] is True and seen['gpu'] == 3 and seen['num_threads'] == 4
    with pytest.raises(SystemExit):
        cli.main(['--output_files', 'x.pt'])              # --input_files is required, as upstream


def test_cpu_route_of_the_many_file_job_never_needs_the_hip_library(tmp_path):
    """`from_files_to_files(gpu=None)` (what `python -m torbi_amd` without --gpu runs) takes its native reader and writer
    from libtorbi_cpu.so (include/torbi_cpu.h: torbi_cpu_read_rows / torbi_cpu_write_files): with the HIP library out
    of reach (TORBI_HIP_LIBRARY names a file that does not exist) the job still runs through the direct file path and
    writes the CPU operator's indices."""
    import subprocess
    import sys
    files = _write_ragged_files(tmp_path, [6, 4, 7, 3, 5, 2, 9])
    outs = [str(tmp_path / f'out{k}.pt') for k in range(len(files))]
    code = ('import sys, torch\n'
            f'sys.path.insert(0, {ROOT!r})\n'
            'import torbi_amd\n'
            'from torbi_amd import _lib, fastio\n'
            'torbi_amd.core.BATCH_SIZE = 3\n'
            f'files, outs = {files!r}, {outs!r}\n'
            'assert isinstance(fastio.open_batches(files, 3, gpu=False), fastio.FileBatches)\n'
            'torbi_amd.from_files_to_files(files, outs, log_probs=True, gpu=None, num_threads=2)\n'
            'assert _lib._LIB is None, "the CPU route loaded libtorbi_hip.so"\n'
            'for f, o in zip(files, outs):\n'
            '    want = torbi_amd.from_probabilities(torch.load(f)[None], log_probs=True, gpu=None)[0]\n'
            '    assert torch.equal(torch.load(o), want)\n'
            'print("ok")\n')
    env = dict(os.environ, TORBI_HIP_LIBRARY=str(tmp_path / 'no_such_library.so'))
    run = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and run.stdout.strip().endswith('ok'), run.stderr[-2000:]


def test_file_batches_producer_lets_go_when_the_consumer_leaves_early(tmp_path):
    """A consumer that stops after the first batch (an exception in its loop, a break) must not leave the reader
    thread blocked on a full queue: sentinel and exceptions are handed over with the same stop-aware loop as batches."""
    import threading
    import time
    from torbi_amd import fastio
    files = _write_ragged_files(tmp_path, [3, 4, 5, 6, 7, 8, 9, 10])
    before = {t.name for t in threading.enumerate()}
    t0 = time.perf_counter()
    for _ in fastio.FileBatches(files, 1, threads=2, pin_memory=False, gpu=False):
        break
    assert time.perf_counter() - t0 < 3.0          # no 5 s join timeout
    time.sleep(0.3)
    assert 'torbi-file-batches' not in {t.name for t in threading.enumerate()} - before


def test_head_route_rejects_a_header_that_promises_more_than_the_file_holds(tmp_path):
    """payload_of_open_file checks the tensor the pickle describes against the file size like payload() checks it
    against the storage record: a container cut inside its payload is UnsupportedFile (the batch then goes through
    torch.load, which raises like the reference would), not a silent read of trailer bytes."""
    from torbi_amd import fastio
    f = str(tmp_path / 'x.pt')
    torch.save(torch.rand(50, 12), f)
    frames, states, start = fastio.payload(f)
    with open(f, 'r+b') as handle:
        handle.truncate(start + 4 * 12 * 20)
    fd = os.open(f, os.O_RDONLY)
    try:
        with pytest.raises(fastio.UnsupportedFile):
            fastio.payload_of_open_file(fd)
    finally:
        os.close(fd)


def test_one_store_keeps_what_is_known_about_tensors_for_as_long_as_they_live():
    """torbi_amd/state.py: scan depths, structure looks, prepared transitions and workspace contents hang off the tensor
    object and its version in ONE store; entries die with their tensors and are never evicted by count -- a hundred other
    matrices do not take a live matrix's notes with them; a write to the tensor starts afresh; reset_path_state() forgets."""
    import gc
    import torch
    import torbi_amd
    from torbi_amd import state, viterbi
    torbi_amd.reset_path_state()
    S = 64
    mine = torch.rand(S, S)
    depth = viterbi._depth_record(mine, S)
    assert depth is not None and viterbi._depth_record(mine, S) is depth
    viterbi._choose_path(mine, mine, 64, S)
    prepared = torbi_amd.core._prepared_transition(mine, True, 'cpu')
    assert torbi_amd.core._prepared_transition(mine, True, 'cpu') is prepared
    others = []
    for k in range(100):                       # a hundred distinct matrices, half of them kept alive
        other = torch.rand(S, S)
        viterbi._depth_record(other, S)
        viterbi._choose_path(other, other, 64, S)
        if k % 2:
            others.append(other)
    del other
    gc.collect()
    assert viterbi._depth_record(mine, S) is depth                # still there
    assert state.size() == 1 + len(others)                        # the dead ones left with their tensors
    assert any(d is depth for d in state.every(('depth', S)))
    mine.add_(1.0)                                                # a new version: nothing carries over
    assert viterbi._depth_record(mine, S) is not depth and ('reach', S) not in state.notes(mine)
    # a workspace remembers which preparation it holds, for its transition's object and version
    ws = torch.empty(16, dtype=torch.uint8)
    assert not viterbi._reusable(ws, mine, (1, 2, 3, 'pruned', 0), True)
    assert viterbi._reusable(ws, mine, (1, 2, 3, 'pruned', 0), True)
    assert not viterbi._reusable(ws, mine, (1, 2, 4, 'pruned', 0), True)
    mine.mul_(2.0)
    assert not viterbi._reusable(ws, mine, (1, 2, 4, 'pruned', 0), True)
    torbi_amd.reset_path_state()
    assert state.size() == 0 and state.peek(mine) is None


def test_timing_scope_like_the_reference():
    """torbi/core.py:200 wraps its operator call in torchutil.time.context('torbi'); torbi/evaluate/core.py:40,118 resets and
    reads the totals.  torbi_amd.timer offers the same three names and from_probabilities opens the same scope."""
    import torch
    import torbi_amd
    torbi_amd.timer.reset()
    assert torbi_amd.timer.results() == {}
    probs = torch.rand(2, 9, 12, generator=torch.Generator().manual_seed(0)).softmax(-1)
    torbi_amd.from_probabilities(probs, gpu=None)
    first = torbi_amd.timer.results()
    assert set(first) == {'torbi'} and first['torbi'] > 0.0
    torbi_amd.from_probabilities(probs, gpu=None)
    with torbi_amd.timer.context('mine'):
        pass
    second = torbi_amd.timer.results()
    assert second['torbi'] > first['torbi'] and 'mine' in second
    torbi_amd.timer.reset()
    assert torbi_amd.timer.results() == {}


def test_design_routing_table_is_what_the_library_answers():
    """DESIGN.md section 4 carries ONE routing table; it is the output of tools/routing_table.py, i.e. of
    torbi_hip_forward_path_on for every (batch, states, requested path) it lists (round-3 review item 7)."""
    import re
    import runpy
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    found = re.search(r'<!-- routing-table:begin -->\n(.*?)\n<!-- routing-table:end -->', text, flags=re.S)
    assert found, 'DESIGN.md lost its routing-table markers'
    table = runpy.run_path(os.path.join(ROOT, 'tools', 'routing_table.py'))['table']()
    assert found.group(1).strip() == table.strip()
    # spot checks of the prose rules under it (256 compute units)
    assert viterbi.forward_path(512, 1440) == 'cluster' and viterbi.forward_path(2049, 1440) == 'resident'
    assert viterbi.forward_path(1, 1440) == 'held' and viterbi.forward_path(8, 1440) == 'rows'
    # current state only, history in HISTORY.md (25 KB in round 4; rounds 5-6 added the band kernels, the NaN rule, roofline.configs)
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 36 * 1024


def test_slab_pool_stops_counting_slabs_that_were_dropped():
    """torbi_amd/slabs.py (round-3 advisor): a slab that is taken and never given back (a batch on a path that does not
    return its buffers, an exception) must not count as out for ever -- the pool would believe it is at its limit and
    allocate a fresh buffer for every batch after."""
    import gc
    from torbi_amd import slabs
    pool = slabs.SlabPool(None)
    first, second = pool.take(1000, limit=2), pool.take(1000, limit=2)
    assert pool._out == 2
    del first
    gc.collect()
    assert pool._out == 1                       # the dropped slab no longer counts
    pool.give(second)
    assert pool._out == 0 and pool.held_bytes() >= 1000
    again = pool.take(500, limit=2)
    assert again is second and pool._out == 1   # ... and what was given back is handed out again
    pool.give(again)
    pool.give(again)                            # (a second give of the same slab does not drive the count below zero)
    assert pool._out == 0


def test_ring_of_staging_chunks_reads_a_batch_in_pieces(tmp_path):
    """torbi_amd/fastio.py::FileBatches.stage_rows (the many-file job's ring of pinned chunks, core.py::_Staging.upload_rows):
    the batch is handed over BEFORE it is read, the stage asks for its rows piece by piece; pieces of any size -- here three
    rows at a time into one small buffer -- assemble the rows `collate` would have built (collate.py:24-31), and the pool
    hands out chunks of exactly the size asked for (the pinned allocator rounds up to powers of two)."""
    import ctypes
    from torbi_amd import fastio, slabs
    gen = torch.Generator().manual_seed(3)
    lengths = [5, 9, 2, 7, 9, 1, 4]
    files = []
    for k, n in enumerate(lengths):
        files.append(tmp_path / f'{k}.pt')
        torch.save(torch.rand(n, 12, generator=gen), files[-1])
    batches = fastio.open_batches(files, 7, pin_memory=True, gpu=False)
    assert batches is not None
    asked = []

    def stage_rows(shape, batch_frames, fill):
        count, longest, states = shape
        out = torch.full(shape, float('nan'))
        piece = torch.empty((3, longest, states))
        for first in range(0, count, 3):
            k = min(3, count - first)
            fill(piece.data_ptr(), first, k)
            asked.append((first, k))
            out[first:first + k] = piece[:k]
        return out
    batches.stage_rows = stage_rows
    (observation, frames, chunks, names), = list(batches)
    assert asked == [(0, 3), (3, 3), (6, 1)] and frames.tolist() == lengths and observation.shape == (7, 9, 12)
    for k, n in enumerate(lengths):
        assert torch.equal(observation[k, :n], torch.load(files[k])) and not observation[k, n:].any()
    # a file that shrinks between the header look and its piece's read: the short read names THAT file
    shrunk = fastio.open_batches(files, 7, pin_memory=True, gpu=False)

    def shrink_then_read(shape, batch_frames, fill):
        piece = torch.empty((3, shape[1], shape[2]))
        fill(piece.data_ptr(), 0, 3)
        with open(files[4], 'r+b') as handle:
            handle.truncate(fastio.payload(files[4])[2] + 4 * 12 * 2)          # two of its nine frames left
        fill(piece.data_ptr(), 3, 3)
    shrunk.stage_rows = shrink_then_read
    with pytest.raises(OSError, match='4.pt'):
        list(shrunk)
    pool = slabs.SlabPool(None)
    chunk = pool.take(1 << 16, limit=2, exact=True)
    assert chunk.numel() == 1 << 16
    pool.give(chunk)
    assert pool.take(1 << 16, limit=2, exact=True) is chunk


def test_band_plan_covers_the_reference_pitch_band_and_says_so_without_a_gpu():
    """torbi_hip_band_members (include/torbi_hip.h): members per 16-item tile of the band kernel, 0 where it does not cover the
    shape.  The plan is host arithmetic (LDS budget of a member: its slab of the band + window + merge buffer within 160 KB;
    every reach within a member's share): the reference's pitch band -- reach 87 of 1440 states, torbi/evaluate/core.py:24-33 --
    fits with eight members per tile, which is what a 512-item batch needs to fill 256 compute units."""
    lib = _lib.load()
    members = lambda items, S, left, right: lib.torbi_hip_band_members(items, S, left, right, 0)
    assert members(512, 1440, 87, 87) == 8                      # 32 tiles x 8 = one workgroup per compute unit
    assert members(4096, 1440, 87, 87) == 1                     # a launch group of eight batches: WHOLE tiles (band_tile_forward.hpp)
    assert members(2064, 1440, 87, 87) == 1 and members(2048, 1440, 87, 87) == 8       # ... once the split form would need five
                                                                # launches of 32 tiles (5 x 7.4 us against 36 us a timestep); four: split
    assert members(4096, 1440, 250, 250) == 1 and members(4096, 1440, 255, 255) == 0   # (the backtrace's window: 512 prev-states)
    assert members(2048, 1440, 250, 250) == 1 and members(2032, 1440, 250, 250) == 0   # no split form for this reach: whole tiles
                                                                                       # from half a tile per unit up, nothing below
    assert members(4096, 1600, 87, 87) == 9                     # more than 24 blocks of 64 next-states: split, 24 tiles per launch
    assert members(16, 1440, 87, 87) == 16                      # one tile: as many as the reach allows (16 x 92 next-states)
    assert members(320, 1440, 87, 87) == 10                     # 20 tiles = 3 per dispatch class: 10 x 3 members on an XCD's 32 units
    assert members(512, 1440, 88, 88) == 9                      # one diagonal more does not fit eight members' LDS: nine
                                                                # (24 tiles per launch: every member of a launch resident)
    assert members(40, 1440, 88, 88) == 16                      # three tiles: as many members as there are compute units for
    assert members(40, 1440, 120, 120) == 12 and members(40, 1440, 121, 121) == 11 and members(40, 1440, 122, 122) == 0
    assert members(512, 1440, 0, 0) == 8 and members(512, 1440, 0, 175) == 8 and members(512, 1440, 176, 0) == 0
    assert members(64, 1442, 87, 87) == 0 and members(64, 32, 3, 3) == 0 and members(64, 4096, 10, 10) == 0
    assert members(64, 1440, 300, 300) == 0
    assert members(0, 1440, 87, 87) == -1 and members(64, 1440, -1, 0) == -1
    from torbi_amd import viterbi
    import torch
    cpu = torch.zeros((1440, 1440))
    assert viterbi.band_reach(cpu, cpu, 1440) is None           # (asked of device tensors only)
