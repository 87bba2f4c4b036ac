"""Parity of the HIP path (through the C ABI, via torbi_amd.decode) with the oracle and the
committed golden vectors.  Bit-exact bar: decoded indices are integers, so every
comparison is array equality (no tolerance)."""
import hashlib
import os

import numpy as np
import pytest
import torch

import oracle
import torbi_amd
from torbi_amd import synth, viterbi
from conftest import SMALL_NAMES, LARGE_NAMES, CachedOracle

oracle = CachedOracle(oracle)        # (one run of the checker per distinct input, not one per forward path)

pytestmark = pytest.mark.gpu


@pytest.fixture
def forward(request):
    """The forward path a path-sensitive test runs under (set process-wide for the test, 'auto' again behind it).  Tests
    marked `@all_paths` run under five: the automatic choice; the dense (max,+) GEMM forced; 'pruned' (the sorted-row scan
    for batches of up to 16 items -- above that it names the time-resident forms); the time-resident kernel forced
    wherever it is supported (64 <= S <= 4096, ANY batch size) with whole 16-item tiles per workgroup; and its cluster form
    (the next-states of a tile split over up to 16 workgroups that exchange their slices of every posterior row inside the
    launch).  `@paths(...)` names fewer where a forced path launches exactly the kernels AUTO launches for every shape of
    the test (fresh tensors, 256 compute units: batches of 17..2047 items over 64..4096 states with a matrix that is not a
    narrow band are 'cluster' under AUTO, and 'pruned' names the same form for them).  A test without
    either mark names its paths itself or never reaches a forward kernel choice: it runs once, under AUTO."""
    name = getattr(request, 'param', 'auto')
    viterbi.set_forward_path(name)
    yield name
    viterbi.set_forward_path('auto')


def paths(*names):
    def mark(fn):
        return pytest.mark.usefixtures('forward')(pytest.mark.parametrize('forward', list(names), indirect=True)(fn))
    return mark


all_paths = paths('auto', 'dense', 'pruned', 'resident', 'cluster')


def gpu_decode(obs, frames, trans, init):
    dev = torch.device('cuda:0')
    out = torbi_amd.decode(torch.as_tensor(obs, dtype=torch.float32).to(dev),
                           torch.as_tensor(np.asarray(frames), dtype=torch.int32).to(dev),
                           torch.as_tensor(trans, dtype=torch.float32).to(dev),
                           torch.as_tensor(init, dtype=torch.float32).to(dev))
    assert out.dtype == torch.int32 and out.device.type == 'cuda'
    return out.cpu().numpy()


def test_extension_is_loaded_and_sees_the_gpu():
    lib = torbi_amd._lib.load()
    assert lib.torbi_hip_device_count() >= 1
    assert 'gfx950' in torch.cuda.get_device_properties(0).gcnArchName


@all_paths
@pytest.mark.parametrize('name', SMALL_NAMES)
def test_golden_small(golden, name):
    obs, frames, trans, init, want = golden.small_case(name)
    assert np.array_equal(gpu_decode(obs, frames, trans, init), want)


@all_paths
@pytest.mark.parametrize('name', LARGE_NAMES)
def test_golden_large(golden, name):
    obs, frames, trans, init, want = golden.large_case(name)
    got = gpu_decode(obs, frames, trans, init)
    assert np.array_equal(got, want)
    assert hashlib.sha256(got.tobytes()).hexdigest() == str(golden.large[name + '/sha256'])


@pytest.mark.parametrize('shape', [(1, 50, 1440), (3, 40, 1440), (24, 30, 360), (33, 20, 257),
                                   (64, 17, 64), (65, 9, 63), (100, 12, 130), (257, 5, 33),
                                   (2, 30, 4096), (40, 6, 2049), (32, 7, 64), (96, 11, 1441),
                                   (512, 3, 100), (128, 5, 4096), (70, 40, 360), (600, 4, 97),
                                   (17, 7, 256), (31, 5, 1440), (24, 6, 1442), (16, 9, 1440), (200, 3, 40),
                                   (256, 4, 360), (300, 3, 1440), (272, 5, 132), (260, 3, 1443)])
@all_paths
@pytest.mark.parametrize('ties', [False, True])
def test_random_shapes_against_oracle(shape, ties):
    B, T, S = shape
    obs, trans, init = synth.problem(B, T, S, seed=B + T + S)
    if ties:   # coarse grid -> many exactly equal candidates, exercises lowest-index rule
        obs, trans, init = np.round(obs / 2), np.round(trans / 2), np.round(init / 2)
    frames = np.clip(synth.lengths(B, 1, T, seed=S), 1, T)
    frames[0] = T
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    assert np.array_equal(gpu_decode(obs, frames, trans, init), want)


@all_paths
@pytest.mark.parametrize('kind', ['banded', 'diagonal', 'blocks', 'dead_rows', 'dead_everything'])
@pytest.mark.parametrize('shape', [(64, 25, 360), (40, 12, 1440), (96, 9, 131)])
def test_dense_path_skips_minus_inf_blocks_exactly(kind, shape):
    """-inf-structured transitions on the large-batch path: chunks of prev-states whose
    transition values are all -inf are skipped (build_chunk_lists_kernel); results must not
    change.  'banded' is the reference's own pitch transition (torbi/evaluate/core.py:24-33)."""
    B, T, S = shape
    obs, trans, init = synth.problem(B, T, S, seed=S + len(kind))
    rng = np.random.default_rng(S)
    if kind == 'banded':
        trans = synth.banded_transition(S, max(3, S // 16))
    elif kind == 'diagonal':
        d = np.full((S, S), -np.inf, np.float32)
        np.fill_diagonal(d, trans.diagonal())
        trans = d
    elif kind == 'blocks':
        mask = rng.random((S // 8 + 1, S // 8 + 1)) < 0.7
        trans = np.where(np.kron(mask, np.ones((8, 8), bool))[:S, :S], -np.inf, trans).astype(np.float32)
    elif kind == 'dead_rows':
        trans = trans.copy()
        trans[rng.random(S) < 0.3] = -np.inf
    else:
        trans = np.full((S, S), -np.inf, np.float32)
    frames = np.clip(synth.lengths(B, 1, T, seed=3), 1, T)
    frames[0] = T
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    assert np.array_equal(gpu_decode(obs, frames, trans, init), want)


@pytest.mark.parametrize('width', [1, 7, 254, 255, 256, 257, 300])
def test_dense_route_backtrace_reads_the_band_only(width):
    """On the dense route the backtrace of a banded matrix reads a row's finite range only (a window of up to 512
    prev-states, lazy::backtrace_ranged_kernel); a matrix with a wider row goes through the whole-row kernel -- decided on
    the device.  Half widths that put the widest row below, on and above the window, rows cut off by the matrix edge, a
    row without any finite entry, ties inside the band."""
    B, T, S = 40, 7, 1440
    obs, trans, init = synth.problem(B, T, S, seed=width)
    obs = np.round(obs * 2) / 2
    trans = np.round(trans * 2) / 2
    idx = np.arange(S)
    trans = np.where(np.abs(idx[:, None] - idx[None, :]) < width, trans, -np.inf).astype(np.float32)
    trans[S // 3] = -np.inf                                       # a next-state nothing leads to
    frames = np.clip(synth.lengths(B, 1, T, seed=2), 1, T).astype(np.int32)
    frames[0] = T
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs.astype(np.float32), frames, trans, init)]
    want = oracle.decode(obs.astype(np.float32), frames, trans, init)
    for _ in range(2):                                            # the second call finds the row ranges in the workspace
        got = torbi_amd.decode(*args, path='dense')
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('shape', [(3, 40, 1440), (70, 9, 360), (5, 1, 64), (2, 33, 4096), (9, 7, 132),
                                   (130, 5, 256), (3, 20, 8), (2, 17, 260), (3, 19, 1000), (2, 9, 2048), (2, 10, 3000)])
@pytest.mark.parametrize('kind', ['plain', 'ties', 'coarse_sums', 'minus_inf'])
def test_uniform_transition_entry_equals_materialised_matrix(shape, kind):
    """torbi_hip_viterbi_decode_uniform vs the oracle run on torch.full((S,S), c), the matrix the
    reference builds for transition=None (torbi/core.py:175-180).  The kernel takes the row maxima off the dependent chain
    (max_i fl(x_i + a) = fl(max_i x_i + a), csrc/uniform_decode.hpp): 'coarse_sums' makes the posteriors ~ -1e7, where
    one ulp is 1.0 and every addition merges many distinct candidates -- the first index among the MERGED maxima has to
    win, as in the reference's scan; 'minus_inf' has -inf candidates and whole -inf rows; lengths straddle the 8-row chunks."""
    import math
    B, T, S = shape
    obs, _, init = synth.problem(B, T, S, seed=S + T)
    if kind == 'ties':
        obs, init = np.round(obs / 4), np.round(init / 4)
    elif kind == 'coarse_sums':
        init = (init - np.float32(1e7)).astype(np.float32)
    elif kind == 'minus_inf':
        rng = np.random.default_rng(S)
        obs = np.where(rng.random(obs.shape) < 0.3, -np.inf, obs).astype(np.float32)
        if T > 4:
            obs[0, 3, :] = -np.inf                                  # a whole row: everything after it is -inf
    c = np.float32(math.log(1. / S))
    frames = np.clip(synth.lengths(B, 1, T, seed=7), 1, T)
    frames[0] = T
    want = oracle.decode(obs, frames, np.full((S, S), c, np.float32), init,
                         num_threads=oracle.max_threads())
    dev = torch.device('cuda:0')
    got = torbi_amd.decode_uniform(torch.tensor(obs, device=dev), torch.tensor(frames, device=dev),
                                   float(c), torch.tensor(init, device=dev))
    assert got.dtype == torch.int32
    assert np.array_equal(got.cpu().numpy(), want)


def test_uniform_entry_at_every_state_count_it_takes():
    """Multiples of 4 up to 516 and the edges of every template instance up to 4096 (a wave owns ceil(S / 256) float4 per
    row): a ragged 3 x 19 batch, scores as they are and as probabilities, against the oracle on the materialised matrix."""
    import math
    dev = torch.device('cuda:0')
    B, T = 3, 19
    frames = np.array([19, 1, 10], np.int32)
    tiny = torch.finfo(torch.float32).tiny
    for S in list(range(4, 520, 4)) + [1020, 1024, 1028, 1536, 1540, 2044, 2048, 2052, 3072, 3076, 4092, 4096]:
        obs, _, init = synth.problem(B, T, S, seed=S)
        c = np.float32(math.log(1. / S))
        full = np.full((S, S), c, np.float32)
        d_obs, d_frames, d_init = torch.tensor(obs, device=dev), torch.tensor(frames, device=dev), torch.tensor(init, device=dev)
        got = torbi_amd.decode_uniform(d_obs, d_frames, float(c), d_init)
        np.testing.assert_array_equal(got.cpu().numpy(), oracle.decode(obs, frames, full, init, num_threads=oracle.max_threads()),
                                      err_msg=f'S = {S}')
        probs = torch.softmax(d_obs, dim=-1)
        scores = torch.log(probs)
        scores.exp_()
        scores += tiny
        scores.log_()
        got = torbi_amd.decode_uniform(probs, d_frames, float(c), d_init, probabilities=True)
        want = oracle.decode(scores.cpu().numpy(), frames, full, init, num_threads=oracle.max_threads())
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f'S = {S}, probabilities')


@pytest.mark.parametrize('shape', [(3, 21, 1440), (40, 9, 360), (2, 5, 4096), (5, 12, 64)])
def test_the_default_call_on_probabilities_is_one_pass_with_the_reference_s_values(shape):
    """from_probabilities(observation) with every default -- probabilities in, no transition, no initial -- runs log(), the
    epsilon round trip and the uniform-transition decode in ONE kernel (torbi_hip_viterbi_decode_uniform_probabilities).
    Against the reference's steps one by one (torbi/core.py:161-206: torch.log, exp_, += tiny, log_, then the operator on
    the materialised log(1/S) matrix -- the oracle): zeros (log -> -inf -> log(tiny)), denormals, ones, values below tiny;
    the caller's tensor is left as it was."""
    import math
    B, T, S = shape
    rng = np.random.default_rng(S + T)
    probs = rng.random((B, T, S)).astype(np.float32)
    probs /= probs.sum(-1, keepdims=True)
    probs[rng.random(probs.shape) < 0.05] = 0.0
    probs[rng.random(probs.shape) < 0.02] = np.float32(1e-42)                  # denormal
    probs[rng.random(probs.shape) < 0.02] = np.float32(3e-39)                  # below tiny, above the denormals' end
    probs[0, 0, :4] = [1.0, 0.0, np.float32(1.17549435e-38), 0.5]
    frames = np.clip(synth.lengths(B, 1, T, seed=3), 1, T)
    frames[0] = T
    dev = torch.device('cuda:0')
    given = torch.tensor(probs, device=dev)
    before = given.clone()
    got = torbi_amd.from_probabilities(given, torch.tensor(frames, device=dev), gpu=0)
    assert torch.equal(given, before)                                          # out of place, like upstream's torch.log
    tiny = torch.finfo(torch.float32).tiny
    scores = torch.log(before)
    scores.exp_()
    scores += tiny
    scores.log_()
    init = np.full((S,), math.log(1. / S + tiny), np.float32)
    trans = np.full((S, S), np.float32(math.log(1. / S)), np.float32)
    want = oracle.decode(scores.cpu().numpy(), frames, trans, init, num_threads=oracle.max_threads())
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    # ... and the kernel's scores are the torch ops' scores bit for bit: decode_uniform on them gives the same indices
    again = torbi_amd.decode_uniform(scores, torch.tensor(frames, device=dev), float(np.float32(math.log(1. / S))),
                                     torch.tensor(init, device=dev))
    assert torch.equal(again, got)


@paths('auto', 'dense', 'resident')
@pytest.mark.parametrize('shape', [(64, 1, 64), (33, 2, 100), (32, 3, 8192), (48, 4, 6148)])
def test_dense_path_edge_shapes(shape):
    """T = 1 (no recurrence step), T = 2, and panels too long for the chunk-list walk (S > 6144)."""
    B, T, S = shape
    obs, trans, init = synth.problem(B, T, S, seed=B + S)
    frames = np.clip(synth.lengths(B, 1, T, seed=5), 1, T)
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    assert np.array_equal(gpu_decode(obs, frames, trans, init), want)


@paths('auto', 'dense')
@pytest.mark.parametrize('S', [2, 3, 4, 5, 8, 9, 16, 17, 31, 32, 33, 63, 64])
def test_one_wavefront_per_sequence_up_to_64_states(S):
    """small_states.hpp: recurrence, byte backpointers and backtrace in one launch.  Lengths around the 4-timestep
    backpointer words and the 64-timestep chunks of the walk back, ties everywhere (scores on a coarse grid: the lowest
    prev-state must win, viterbi.cpp:94-100), -inf rows and columns, and the final posterior rows where
    torbi_hip_read_posterior looks for them."""
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(S)
    for B, T in [(1, 1), (3, 2), (5, 5), (4, 9), (70, 67), (6, 130), (2, 517)]:
        obs = -rng.integers(0, 6, size=(B, T, S)).astype(np.float32)          # coarse grid: many exact ties
        trans = -rng.integers(0, 5, size=(S, S)).astype(np.float32)
        init = -rng.integers(0, 3, size=(S,)).astype(np.float32)
        trans[rng.random((S, S)) < 0.2] = -np.inf
        obs[rng.random((B, T, S)) < 0.05] = -np.inf
        if S > 2:
            trans[:, S - 1] = -np.inf                                          # a state nobody can come from
        frames = np.array([T, 1, max(T - 1, 1), max(T - 3, 1), max(T - 4, 1), max(T // 2, 1), max(T - 63, 1),
                           max(T - 64, 1), max(T - 65, 1)], np.int32)
        frames = np.resize(frames, B).astype(np.int32)
        want, post = oracle.decode(obs, frames, trans, init, return_posterior=True)
        ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
        profile = []
        got = torbi_amd.decode(torch.tensor(obs, device=dev), torch.tensor(frames, device=dev), torch.tensor(trans, device=dev),
                               torch.tensor(init, device=dev), workspace=ws, _profile=profile)
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f'{B} x {T} x {S}')
        if viterbi.forward_path(B, S) == 'small':
            assert viterbi.ROUTES[int(profile[3])] == 'small' and profile[2] == 1       # ONE launch
            assert viterbi.last_forward_kernel().startswith('small::decode_kernel<')
        back = viterbi.read_posterior(ws, torch.tensor(frames), B, T, S).cpu().numpy()
        assert np.array_equal(back.view(np.uint32), post.view(np.uint32)), (B, T, S)
    # random real-valued scores too (no ties): the usual synthetic problem, ragged
    B, T = 33, 200
    obs, trans, init = synth.problem(B, T, S, seed=S)
    frames = np.clip(synth.lengths(B, 1, T, seed=S), 1, T)
    np.testing.assert_array_equal(gpu_decode(obs, frames, trans, init), oracle.decode(obs, frames, trans, init))


@paths('auto', 'dense')
@pytest.mark.parametrize('form', ['value', 'pairs'])
@pytest.mark.parametrize('S', [65, 80, 81, 100, 128, 129, 160, 161, 192, 193, 224, 225, 255, 256])
def test_one_workgroup_per_sequence_up_to_256_states(S, form, monkeypatch):
    """small_states.hpp, block_value_kernel: the matrix in the registers of one compute unit, the prev-states in one or
    two ranges, four running maxima per lane, posterior rows kept; the backtrace a launch pair of its own -- in speculative
    segments where the matrix is 16-byte aligned and S % 4 == 0, whole paths otherwise.  One sequence per workgroup and two
    (`pairs`).  Ties everywhere (coarse grid), -inf rows / columns / observations, ragged lengths around the 4-timestep
    observation prefetch, the final posterior rows."""
    monkeypatch.setenv('TORBI_HIP_BLOCK_PAIRS', '1' if form == 'pairs' else '0')       # (two sequences per workgroup)
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(S)
    for B, T in [(1, 1), (3, 2), (5, 5), (9, 67), (3, 130), (2, 261)]:
        obs = -rng.integers(0, 6, size=(B, T, S)).astype(np.float32)
        trans = -rng.integers(0, 5, size=(S, S)).astype(np.float32)
        init = -rng.integers(0, 3, size=(S,)).astype(np.float32)
        trans[rng.random((S, S)) < 0.2] = -np.inf
        obs[rng.random((B, T, S)) < 0.05] = -np.inf
        trans[:, S - 1] = -np.inf
        trans[S // 2, :] = -np.inf                                             # a state nobody can reach
        frames = np.resize(np.array([T, 1, max(T - 1, 1), max(T - 3, 1), max(T - 4, 1), max(T // 2, 1), max(T - 63, 1),
                                     max(T - 64, 1), max(T - 65, 1)], np.int32), B).astype(np.int32)
        want, post = oracle.decode(obs, frames, trans, init, return_posterior=True)
        ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
        profile = []
        got = torbi_amd.decode(torch.tensor(obs, device=dev), torch.tensor(frames, device=dev), torch.tensor(trans, device=dev),
                               torch.tensor(init, device=dev), workspace=ws, _profile=profile)
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f'{B} x {T} x {S}')
        if viterbi.forward_path(B, S) == 'small':
            assert viterbi.ROUTES[int(profile[3])] == 'small' and profile[2] == 2
            assert viterbi.last_forward_kernel().startswith('small::block_value_kernel<')
        back = viterbi.read_posterior(ws, torch.tensor(frames), B, T, S).cpu().numpy()
        assert np.array_equal(back.view(np.uint32), post.view(np.uint32)), (B, T, S)
    B, T = 21, 150
    obs, trans, init = synth.problem(B, T, S, seed=S)
    frames = np.clip(synth.lengths(B, 1, T, seed=S), 1, T)
    np.testing.assert_array_equal(gpu_decode(obs, frames, trans, init), oracle.decode(obs, frames, trans, init))


def test_every_state_count_up_to_256():
    """Every S in 2..256 once (each template instance of small_states.hpp at its edges: padded state counts, the pieces
    of the prev-states, workgroups of 4 / 9 / 16 waves), a ragged 5 x 23 batch of coarse-grid scores with -inf entries."""
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(256)
    B, T = 5, 23
    frames = np.array([23, 1, 22, 9, 17], np.int32)
    for S in range(2, 257):
        obs = -rng.integers(0, 7, size=(B, T, S)).astype(np.float32)
        trans = -rng.integers(0, 6, size=(S, S)).astype(np.float32)
        trans[rng.random((S, S)) < 0.15] = -np.inf
        init = -rng.integers(0, 3, size=(S,)).astype(np.float32)
        want = oracle.decode(obs, frames, trans, init)
        got = torbi_amd.decode(torch.tensor(obs, device=dev), torch.tensor(frames, device=dev), torch.tensor(trans, device=dev),
                               torch.tensor(init, device=dev))
        assert viterbi.forward_path(B, S) == 'small'
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f'S = {S}')


@all_paths
def test_forward_path_selection(forward):
    assert viterbi.forward_path(4, 1440) == {'resident': 'resident', 'cluster': 'cluster', 'pruned': 'rows'}.get(forward, 'generic')
    assert viterbi.forward_path(3, 1440) == {'resident': 'resident', 'cluster': 'cluster', 'pruned': 'rows',
                                             'auto': 'held'}.get(forward, 'generic')   # AUTO: held matrix up to 3 items
    assert viterbi.forward_path(16, 1440, path='held') == 'held' and viterbi.forward_path(17, 1440, path='held') == 'cluster'
    assert viterbi.forward_path(1, 4096, path='auto') == 'held' and viterbi.forward_path(1, 4100, path='auto') == 'generic'
    assert viterbi.forward_path(12, 1440) == {'resident': 'resident', 'cluster': 'cluster', 'dense': 'generic'}.get(forward, 'rows')
    # (AUTO up to 16 items: the sorted-row scan while items x states <= 12 x 1440, one tile split over sixteen members beyond)
    assert viterbi.forward_path(16, 1440) == {'resident': 'resident', 'cluster': 'cluster', 'dense': 'generic',
                                              'auto': 'cluster'}.get(forward, 'rows')
    assert viterbi.forward_path(2, 4096) == {'dense': 'generic', 'resident': 'resident', 'cluster': 'cluster',
                                             'auto': 'held'}.get(forward, 'rows')
    assert viterbi.forward_path(9, 4096) == {'dense': 'generic', 'resident': 'resident', 'cluster': 'cluster',
                                             'auto': 'cluster'}.get(forward, 'rows')
    # up to 64 states: one wavefront per sequence whatever the batch (a named path that covers the shape keeps it; one that
    # does not falls back as AUTO would; DENSE named below 32 items or 64 states: the generic kernels)
    assert viterbi.forward_path(4, 40) == ('generic' if forward == 'dense' else 'small')
    assert viterbi.forward_path(1, 40) == ('generic' if forward == 'dense' else 'small')
    assert viterbi.forward_path(512, 256, path='auto') == 'small' and viterbi.forward_path(512, 257, path='auto') == 'cluster'
    assert viterbi.forward_path(512, 64, path='resident') == 'resident' and viterbi.forward_path(1, 1, path='auto') == 'held'
    assert viterbi.forward_path(4, 4100) == 'generic'
    assert viterbi.forward_path(128, 4096) == {'dense': 'dense', 'resident': 'resident'}.get(forward, 'cluster')   # 8-item tiles
    assert viterbi.forward_path(128, 4100) == 'dense'              # posterior tile does not fit the LDS
    assert viterbi.forward_path(512, 1440) == {'dense': 'dense', 'resident': 'resident'}.get(forward, 'cluster')
    assert viterbi.forward_path(64, 130) == {'dense': 'dense', 'resident': 'resident', 'cluster': 'cluster', 'pruned': 'cluster'}.get(forward, 'small')
    assert viterbi.forward_path(64, 300) == {'dense': 'dense', 'resident': 'resident'}.get(forward, 'cluster')
    # AUTO: a batch that gives more than half the compute units a 16-item workgroup is decoded time-resident with whole
    # tiles per workgroup, a smaller one of more than 16 items (up to 2048 states) in clusters of workgroups per tile
    cus = viterbi.compute_units('cuda:0')
    assert viterbi.forward_path(16 * cus, 1440, path='auto') == 'resident'
    assert viterbi.forward_path(16 * cus, 1440, path='cluster') == 'resident'     # nothing to split
    assert viterbi.forward_path(8 * cus, 1440, path='auto') == 'cluster'          # half the chip: two workgroups per tile
    assert viterbi.forward_path(8 * cus, 2052, path='auto') == 'resident'         # 8-item tiles above 2048 states
    assert viterbi.forward_path(8 * cus, 4100, path='auto') == 'dense'            # the posterior tile does not fit the LDS
    assert viterbi.forward_path(3 * cus, 1440, path='auto') == 'cluster'
    assert viterbi.forward_path(2 * cus, 1440, path='auto') == 'cluster'
    assert viterbi.forward_path(17, 1440, path='auto') == 'cluster'
    assert viterbi.forward_path(16, 1440, path='auto') == 'cluster' and viterbi.forward_path(12, 1440, path='auto') == 'rows'
    assert viterbi.forward_path(16, 512, path='auto') == 'rows' and viterbi.forward_path(12, 2048, path='auto') == 'cluster'
    assert viterbi.forward_path(128, 4096, path='auto') == 'cluster'              # 8-item tiles: 16 tiles x 16 members
    assert viterbi.forward_path(128, 4100, path='auto') == 'dense' and viterbi.forward_path(40, 40, path='auto') == 'small'
    assert viterbi.forward_path(2 * cus, 1440, path='pruned') == 'cluster'       # (its per-timestep tile kernel is gone)
    assert viterbi.forward_path(16 * cus, 1440, path='pruned') == 'resident' and viterbi.forward_path(9, 1440, path='pruned') == 'rows'
    # the path travels with the call: naming one never changes the process default
    assert viterbi.forward_path(512, 1440, path='dense') == 'dense'
    assert viterbi.workspace_bytes(512, 500, 1440) >= 512 * 500 * 1440 * 4


@pytest.mark.parametrize('shape', [(1, 50, 1440), (1, 2, 3), (2, 1, 64), (3, 17, 100), (4, 9, 511), (2, 12, 512), (5, 8, 513),
                                   (16, 6, 1440), (3, 5, 2048), (7, 11, 1027), (1, 300, 360), (2, 5, 2052), (3, 4, 3072),
                                   (2, 7, 3100), (1, 6, 4096), (16, 3, 4096)])
@pytest.mark.parametrize('kind', ['random', 'ties', 'minus_inf'])
def test_held_matrix_kernel_equals_the_oracle(shape, kind):
    """The one-launch forward pass for a handful of sequences (csrc/held_matrix_forward.hpp: the matrix in registers
    across the chip, posterior rows exchanged as {value, timestep} words): ragged lengths, heavy ties (the first
    maximum must win through the (value, index) folds), -inf transitions and whole -inf observation rows, every
    prev-states-per-thread variant (S <= 512, 1024, 1536, 2048 with 8 rows per workgroup; <= 3072, 4096 with 16) and state
    counts that leave threads and rows idle."""
    B, T, S = shape
    obs, trans, init = synth.problem(B, T, S, seed=S + B)
    rng = np.random.default_rng(B * 1000 + S)
    if kind == 'ties':
        obs = np.round(obs / 4).astype(np.float32)
        trans = np.round(trans / 4).astype(np.float32)
        init = np.round(init / 4).astype(np.float32)
    elif kind == 'minus_inf':
        trans = trans.copy()
        trans[rng.random((S, S)) < 0.6] = -np.inf
        trans[rng.integers(0, S)] = -np.inf                       # a next-state nothing leads to
        obs = obs.copy()
        if T > 2:
            obs[0, T // 2] = -np.inf                              # a whole frame without support
    frames = np.clip(synth.lengths(B, 1, T, seed=3), 1, T).astype(np.int32)
    frames[0] = T
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    want, want_post = oracle.decode(obs, frames, trans, init, return_posterior=True)
    prof = []
    for _ in range(2):                                            # the second decode finds the first one's words in place
        got = torbi_amd.decode(*args, workspace=space, path='held', _profile=prof)
        assert int(prof[3]) == 6 and int(prof[2]) == 1            # route 'held', ONE forward launch
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    post = viterbi.read_posterior(space, args[1], B, T, S).cpu().numpy()
    assert np.array_equal(post.view(np.uint32), want_post.view(np.uint32))


@pytest.mark.parametrize('path', ['dense', 'held'])
@pytest.mark.parametrize('T', [129, 160, 161, 257, 500])
def test_parallel_chase_of_long_sequences(T, path):
    """The backpointer chase of a handful of long sequences runs as three short launches (chunk maps for every state,
    boundary states, chunk interiors; torbi_hip.hip chase_*_kernel) from 129 frames on: lengths on and next to every
    chunk boundary, length 1 and 2, under the per-timestep trellis kernels and the held-matrix kernel."""
    B, S = 12, 96
    obs, trans, init = synth.problem(B, T, S, seed=T)
    frames = np.array([T, 1, 2, 31, 32, 33, 64, 65, T - 1, 96, 97, (T // 32) * 32], dtype=np.int32)
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    want = oracle.decode(obs, frames, trans, init)
    got = torbi_amd.decode(*args, path=path)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('shape', [(2, 3000, 360), (1, 1500, 1440), (3, 700, 2050)])
def test_held_matrix_kernel_over_many_timesteps(shape):
    """Thousands of hand-offs in one launch (two parities of {value, timestep} words, every workgroup waiting for all the
    others every timestep): ragged lengths, AUTO's own choice of the path, indices against the oracle."""
    B, T, S = shape
    obs, trans, init = synth.problem(B, T, S, seed=T)
    frames = np.clip(synth.lengths(B, T // 2, T, seed=1), 1, T).astype(np.int32)
    frames[0] = T
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    prof = []
    got = torbi_amd.decode(*args, _profile=prof)
    assert int(prof[3]) == 6 and int(prof[2]) == 1
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('shape', [(1, 40, 1440), (3, 25, 360), (2, 6, 4096), (5, 9, 130)])
def test_held_launch_that_cannot_complete_is_repaired(shape, monkeypatch):
    """A held-matrix launch needs all its workgroups resident at once; when they are not (several such launches from
    different streams on a full device) its polls run out, the workgroups go on without waiting and `repair_kernel` decodes
    the sequences again.  Forced here with a poll limit of 0: every wait gives up at once (counted in the statistics),
    the indices and the final posterior rows are still the oracle's; without the limit nothing gives up."""
    B, T, S = shape
    obs, trans, init = synth.problem(B, T, S, seed=77)
    frames = np.clip(synth.lengths(B, 1, T, seed=5), 1, T).astype(np.int32)
    frames[0] = T
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    want, want_post = oracle.decode(obs, frames, trans, init, return_posterior=True)
    for limit, gave_up in (('0', True), (None, False), ('0', True)):
        if limit is None:
            monkeypatch.delenv('TORBI_HIP_HELD_SPIN_LIMIT', raising=False)
        else:
            monkeypatch.setenv('TORBI_HIP_HELD_SPIN_LIMIT', limit)
        got = torbi_amd.decode(*args, workspace=space, path='held')
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        post = viterbi.read_posterior(space, args[1], B, T, S).cpu().numpy()
        assert np.array_equal(post.view(np.uint32), want_post.view(np.uint32))
        stats = viterbi.scan_stats(space, B, T, S).cpu()
        assert (int(stats[127]) > 0) == gave_up, (limit, int(stats[127]))


@pytest.mark.parametrize('shape', [(40, 12, 360), (17, 9, 1440), (130, 7, 724), (20, 6, 2052)])
def test_cluster_that_gives_up_waiting_is_repaired(shape, monkeypatch):
    """A cluster launch whose members cannot all arrive in time (a device shared with work that holds the compute units)
    used to return incomplete histories (round-3 advisor).  Now a member that gives up flags its tile and the launch
    behind the cluster launch decodes the flagged tiles again, whole.  Forced with a wait budget of 0 (every poll that
    fails gives up): indices and final posterior rows are the oracle's, the give-ups are counted; without the limit
    nothing gives up."""
    B, T, S = shape
    obs, trans, init = synth.problem(B, T, S, seed=31)
    frames = np.clip(synth.lengths(B, 1, T, seed=6), 1, T).astype(np.int32)
    frames[0] = T
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    want, want_post = oracle.decode(obs, frames, trans, init, return_posterior=True)
    for limit, gave_up in (('0', True), (None, False), ('0', True)):
        if limit is None:
            monkeypatch.delenv('TORBI_HIP_CLUSTER_WAIT_US', raising=False)
        else:
            monkeypatch.setenv('TORBI_HIP_CLUSTER_WAIT_US', limit)
        got = torbi_amd.decode(*args, workspace=space, path='cluster')
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        post = viterbi.read_posterior(space, args[1], B, T, S).cpu().numpy()
        assert np.array_equal(post.view(np.uint32), want_post.view(np.uint32))
        stats = viterbi.scan_stats(space, B, T, S, path='resident').cpu()
        assert (int(stats[127]) > 0) == gave_up, (limit, int(stats[127]))


def test_a_single_sequence_beside_a_busy_stream_keeps_away_from_the_held_kernel():
    """Round-3 review item 5: the held-matrix kernel needs all its workgroups resident at once.  A B = 1 AUTO decode
    issued while a full-chip time-resident launch group occupies ANOTHER stream must not sit out a wait budget and the
    slow repair: AUTO sees the other stream's work in flight (the library marks the end of every decode with an event)
    and takes the per-timestep kernels, which queue behind it.  Asserted: oracle equality, no give-ups, wall time of the
    pair <= the launch group alone + 5 ms; and once the device is idle again the same call is back on the held kernel."""
    import time
    dev = torch.device('cuda:0')
    S, T = 1440, 200
    trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
    init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
    big = [viterbi.fill_synthetic((512, T, S), synth.STREAM_OBSERVATION, seed=k, device=dev) for k in range(8)]
    big_frames = [torch.full((512,), T, dtype=torch.int32, device=dev)] * 8
    spaces = [torch.empty(viterbi.workspace_bytes(512, T, S), dtype=torch.uint8, device=dev) for _ in range(8)]
    one = viterbi.fill_synthetic((1, T, S), synth.STREAM_OBSERVATION, seed=99, device=dev)
    one_frames = torch.full((1,), T, dtype=torch.int32, device=dev)
    one_space = torch.empty(viterbi.workspace_bytes(1, T, S), dtype=torch.uint8, device=dev)
    want = oracle.decode(one.cpu().numpy(), [T], trans.cpu().numpy(), init.cpu().numpy(), num_threads=oracle.max_threads())
    side = torch.cuda.Stream(device=dev)

    def group():
        with torch.cuda.stream(side):
            viterbi.decode_batches(big, big_frames, trans, init, workspaces=spaces, path='resident')

    # idle device: AUTO takes the held kernel
    prof = []
    torbi_amd.decode(one, one_frames, trans, init, workspace=one_space, _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == 'held'
    group()
    torch.cuda.synchronize()                       # (warm: code objects, LDS grants)
    timings = []
    for attempt in range(3):                       # wall times on a shared box: the bound must hold in one of three rounds
        t0 = time.perf_counter()
        group()
        torch.cuda.synchronize()
        alone = time.perf_counter() - t0
        t0 = time.perf_counter()
        group()
        got = torbi_amd.decode(one, one_frames, trans, init, workspace=one_space)  # default stream, the group in flight
        torch.cuda.synchronize()
        both = time.perf_counter() - t0
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        stats = viterbi.scan_stats(one_space, 1, T, S).cpu()
        assert int(stats[127]) == 0
        timings.append((alone, both))
        if both <= alone + 5e-3:
            break
    assert any(both <= alone + 5e-3 for alone, both in timings), timings
    # what ran: the route record in the workspace says per-timestep kernels (0 generic / 4 rows), not held (6)
    prof = []
    group()
    torbi_amd.decode(one, one_frames, trans, init, workspace=one_space, _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] in ('generic', 'rows'), prof[3]
    torch.cuda.synchronize()
    torbi_amd.decode(one, one_frames, trans, init, workspace=one_space, _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == 'held'


def test_concurrent_held_launches_from_several_streams():
    """Six host threads, six streams, one 4096-state sequence each on the held-matrix kernel -- 256 workgroups of 1024
    threads per launch, one per compute unit, so the launches cannot all be resident together.  Whatever the dispatcher
    does (one after the other, or interleaved until polls run out and the repair kernel steps in), every result is the
    oracle's."""
    import threading
    dev = torch.device('cuda:0')
    T, S, n = 24, 4096, 6
    _, trans, init = synth.problem(1, 1, S, seed=9)
    d_trans, d_init = torch.as_tensor(trans).to(dev), torch.as_tensor(init).to(dev)
    jobs = []
    for k in range(n):
        obs = synth.scores(synth.STREAM_OBSERVATION, (1, T, S), seed=300 + k)
        frames = np.array([T - k], dtype=np.int32)
        jobs.append((obs, frames, oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())))
    results, errors = [None] * n, []

    def work(k):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                d_obs, d_frames = torch.as_tensor(jobs[k][0]).to(dev), torch.as_tensor(jobs[k][1]).to(dev)
                space = torch.empty(viterbi.workspace_bytes(1, T, S), dtype=torch.uint8, device=dev)
                for _ in range(3):
                    got = torbi_amd.decode(d_obs, d_frames, d_trans, d_init, workspace=space, path='held')
                stream.synchronize()
                results[k] = got.cpu().numpy()
        except Exception as exc:                      # surfaced below: a thread must not fail silently
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for k in range(n):
        np.testing.assert_array_equal(results[k], jobs[k][2], err_msg=f'stream {k}')


def test_auto_looks_at_the_transition_once_per_tensor_version():
    """The Python layer's look at the transition decides where ONE batch of more than 16 items goes: the band kernel for a
    matrix that is -inf outside a band it covers (csrc/band_forward.hpp; dense + -inf skipping until round 4), clusters
    otherwise (16- and 8-item tiles alike) -- once per tensor version."""
    dev = torch.device('cuda:0')
    B, T, S = 64, 6, 360
    obs, trans, init = synth.problem(B, T, S, seed=3)
    frames = np.full(B, T, dtype=np.int32)
    args = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    prof = []
    torbi_amd.decode(*args, _profile=prof)
    assert int(prof[3]) == 5                                         # dense random matrix: time-resident clusters
    band = torch.as_tensor(synth.banded_transition(S, 20.0)).to(dev)
    torbi_amd.decode(args[0], args[1], band, args[3], _profile=prof)
    assert int(prof[3]) == 8                                         # narrow band: the band kernel
    B, T, S = 40, 3, 2064
    obs, trans, init = synth.problem(B, T, S, seed=3)
    frames = np.full(B, T, dtype=np.int32)
    args = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    torbi_amd.decode(*args, _profile=prof)
    assert int(prof[3]) == 5                                         # dense random matrix: clusters on 8-item tiles
    band = torch.as_tensor(synth.banded_transition(S, 20.0)).to(dev)
    torbi_amd.decode(args[0], args[1], band, args[3], _profile=prof)
    assert int(prof[3]) == 8                                         # narrow band: the band kernel
    band.fill_(-1.0)                                                 # same storage, new version: ONE value everywhere is no band
    torbi_amd.decode(args[0], args[1], band, args[3], _profile=prof)
    assert int(prof[3]) == 5


@pytest.mark.parametrize('kind', ['flat', 'nearly_flat', 'peaked', 'anti', 'two_level'])
@pytest.mark.parametrize('shape', [(32, 12, 64), (33, 9, 132), (48, 6, 1444), (40, 5, 2048), (64, 10, 360),
                                   (36, 4, 2052), (32, 3, 4096), (256, 4, 724), (270, 3, 1440)])
@paths('auto', 'dense', 'resident')
def test_pruned_path_adversarial_inputs(kind, shape):
    """Inputs chosen against the pruning bound: rows without spread (nothing can be pruned: the scan runs to
    the end of every list), posteriors with a few dominant peaks (the explicit seeds carry the maximum),
    transitions anti-correlated with the observations, and heavy ties."""
    B, T, S = shape
    obs, trans, init = synth.problem(B, T, S, seed=41)
    rng = np.random.default_rng(S + T)
    if kind == 'flat':
        trans = np.full((S, S), np.float32(-1.25))
    elif kind == 'nearly_flat':
        trans = (trans * np.float32(2 ** -12)).astype(np.float32)
    elif kind == 'peaked':
        peaks = rng.integers(0, S, size=(B, T, 2))
        obs = (obs - np.float32(40.0)).astype(np.float32)
        for k in range(2):
            np.put_along_axis(obs, peaks[:, :, k:k + 1], np.float32(-0.5 * k), axis=2)
    elif kind == 'anti':
        # large transitions exactly where the first observations are small
        trans = (-obs[0, 0][None, :] - np.float32(16.0) + trans * np.float32(2 ** -6)).astype(np.float32)
    elif kind == 'two_level':
        trans = np.where(rng.random((S, S)) < 0.1, np.float32(-1.0), np.float32(-3.0)).astype(np.float32)
        obs = np.round(obs).astype(np.float32)
    frames = synth.lengths(B, 1, T, seed=S)
    want = oracle.decode(obs, frames, trans, init, mode=1)
    np.testing.assert_array_equal(gpu_decode(obs, frames, trans, init), want)


@all_paths
@pytest.mark.parametrize('seed', [11, 12, 13])
def test_randomised_shapes_lengths_and_structures(seed):
    """A slice of tools/stress.py (1200 cases x 3 paths clean on the box): random batch/state counts around the
    path thresholds, ragged lengths, -inf densities, bands, heavy ties and unprunable matrices."""
    rng = np.random.default_rng(seed)
    for _ in range(12):
        S = int(rng.choice([rng.integers(16, 80), rng.integers(16, 560) * 4, rng.integers(64, 2200)]))
        B = int(rng.choice([rng.integers(1, 20), rng.integers(17, 70), rng.integers(60, 160)]))
        T = int(rng.integers(1, 9))
        if B * T * S * S > 3e9:
            B = max(1, int(3e9 / (T * S * S)))
        obs, trans, init = synth.problem(B, T, S, seed=int(rng.integers(1 << 30)))
        kind = int(rng.integers(5))
        if kind == 1:
            obs = np.round(obs / 4) * 4
            trans = np.round(trans / 8) * 8
        elif kind == 2:
            trans = np.where(rng.random((S, S)) < rng.choice([0.3, 0.9, 0.99]), -np.inf, trans)
        elif kind == 3:
            reach = rng.integers(1, max(2, S // 4))
            trans = np.where(np.abs(np.arange(S)[:, None] - np.arange(S)[None, :]) > reach, -np.inf, trans)
        elif kind == 4:
            trans = trans * np.float32(2.0 ** -int(rng.integers(8, 20)))
        obs, trans, init = (np.ascontiguousarray(x, dtype=np.float32) for x in (obs, trans, init))
        frames = rng.integers(1, T + 1, size=B).astype(np.int32)
        want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
        np.testing.assert_array_equal(gpu_decode(obs, frames, trans, init), want,
                                      err_msg=f'B={B} T={T} S={S} kind={kind}')


@all_paths
@pytest.mark.parametrize('B', [2, 40])
def test_out_of_range_lengths_are_clamped(B):
    """batch_frames outside [1, T] is clamped on the device (the reference reads out of bounds
    for 0, viterbi.cpp:153): 0 and negatives behave as 1, values > T as T."""
    T, S = 9, 72
    obs, trans, init = synth.problem(B, T, S, seed=B)
    frames = np.full((B,), T, np.int32)
    frames[0], frames[1] = 0, T + 7
    clamped = np.clip(frames, 1, T)
    want = oracle.decode(obs, clamped, trans, init)
    assert np.array_equal(gpu_decode(obs, frames, trans, init), want)


@all_paths
def test_non_contiguous_inputs_are_accepted():
    """the reference calls .contiguous() inside the op (viterbi.cu:325-328)"""
    dev = torch.device('cuda:0')
    B, T, S = 34, 6, 80
    obs, trans, init = synth.problem(B, T, 2 * S, seed=1)
    o = torch.tensor(obs, device=dev)[:, :, ::2]
    tr = torch.tensor(synth.scores(2, (2 * S, S), seed=1), device=dev)[::2]
    i = torch.tensor(init, device=dev)[::2]
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    got = torbi_amd.decode(o, frames, tr, i).cpu().numpy()
    want = oracle.decode(o.cpu().numpy(), frames.cpu().numpy(), tr.cpu().numpy(), i.cpu().numpy())
    assert np.array_equal(got, want)


@all_paths
@pytest.mark.parametrize('B', [48, 12])
def test_preparation_reuse_follows_the_transition_and_the_shape(B):
    """decode(workspace=..., reuse_preparation=True) skips the per-transition preparation only when the
    workspace's previous decode had the same shape, path and transition version (include/torbi_hip.h,
    TORBI_HIP_REUSE_TRANSITION); everything else rebuilds it.  (B = 12: the small-batch kernels.)"""
    dev = torch.device('cuda:0')
    T, S = 7, 360
    obs, trans, init = synth.problem(B, T, S, seed=5)
    obs2 = synth.problem(B, T, S, seed=6)[0]
    frames = synth.lengths(B, 1, T, seed=2)
    d = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    ws = torch.empty(viterbi.workspace_bytes(64, 9, 512), dtype=torch.uint8, device=dev)
    want1 = oracle.decode(obs, frames, trans, init)
    want2 = oracle.decode(obs2, frames, trans, init)
    for _ in range(2):
        np.testing.assert_array_equal(torbi_amd.decode(*d, workspace=ws, reuse_preparation=True).cpu().numpy(), want1)
    got = torbi_amd.decode(torch.as_tensor(obs2).to(dev), d[1], d[2], d[3], workspace=ws, reuse_preparation=True)
    np.testing.assert_array_equal(got.cpu().numpy(), want2)            # same matrix, new observations: reused
    d[2].mul_(0.5)                                                     # new version of the matrix: rebuilt
    want3 = oracle.decode(obs, frames, (trans * np.float32(0.5)).astype(np.float32), init)
    np.testing.assert_array_equal(torbi_amd.decode(*d, workspace=ws, reuse_preparation=True).cpu().numpy(), want3)
    # another shape in between invalidates what the workspace holds
    o4, t4, i4 = synth.problem(40, 5, 512, seed=9)
    f4 = np.full(40, 5, dtype=np.int32)
    d4 = [torch.as_tensor(x).to(dev) for x in (o4, f4, t4, i4)]
    np.testing.assert_array_equal(torbi_amd.decode(*d4, workspace=ws, reuse_preparation=True).cpu().numpy(),
                                  oracle.decode(o4, f4, t4, i4))
    np.testing.assert_array_equal(torbi_amd.decode(*d, workspace=ws, reuse_preparation=True).cpu().numpy(), want3)
    lib = torbi_amd._lib.load()
    assert lib.torbi_hip_viterbi_decode_ex(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                           got.data_ptr(), ws.data_ptr(), ws.numel(), B, T, S, 0, None, 4) == -1


@all_paths
def test_calls_without_a_workspace_keep_the_preparation_with_the_matrix():
    """decode() / decode_batches() allocate their scratch per call like the reference's operator (viterbi.cu:331-336); the
    time-resident routes' per-transition preparation then lives with the transition tensor (include/torbi_hip.h,
    torbi_hip_viterbi_decode_batches_prepared): built by the first call, found by the next ones -- on any stream --,
    dropped when the matrix changes, never trusted by a route that does not use it."""
    from torbi_amd import state
    dev = torch.device('cuda:0')
    B, T, S = 40, 9, 360
    obs, trans, init = synth.problem(B, T, S, seed=15)
    obs2 = synth.problem(B, T, S, seed=16)[0]
    frames = synth.lengths(B, 1, T, seed=3)
    d = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    resident = viterbi.forward_path(B, S) in viterbi.TIME_RESIDENT           # (not under the forced dense path)
    small = [torch.as_tensor(x).to(dev) for x in (obs[:4], frames[:4])]
    # a route that keeps no such preparation first: nothing is marked as filled, its results are right
    np.testing.assert_array_equal(torbi_amd.decode(small[0], small[1], d[2], d[3]).cpu().numpy(),
                                  oracle.decode(obs[:4], frames[:4], trans, init))
    if viterbi.forward_path(4, S) not in viterbi.TIME_RESIDENT:
        assert not any(k[0] == 'preparation' for k in (state.peek(d[2]) or {}) if isinstance(k, tuple))
    want1 = oracle.decode(obs, frames, trans, init)
    want2 = oracle.decode(obs2, frames, trans, init)
    np.testing.assert_array_equal(torbi_amd.decode(*d).cpu().numpy(), want1)
    if resident:
        kept = state.peek(d[2])[('preparation', S, 0)]
        assert kept.filled is not None and kept.buffer.numel() >= torbi_amd._lib.load().torbi_hip_preparation_bytes(S)
        pointer = kept.buffer.data_ptr()
    profile = []
    np.testing.assert_array_equal(torbi_amd.decode(torch.as_tensor(obs2).to(dev), *d[1:], _profile=profile).cpu().numpy(), want2)
    if resident:
        assert state.peek(d[2])[('preparation', S, 0)].buffer.data_ptr() == pointer
        assert profile[4] < 0.05, profile                                  # ms of preparation: nothing was rebuilt
    other = torch.cuda.Stream(dev)
    other.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(other):                                        # another stream: ordered behind the filling call
        got = torbi_amd.decode(*d)
    other.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), want1)
    # a group of batches, and the small route between two uses
    outs = torbi_amd.decode_batches([d[0], small[0]], [d[1], small[1]], d[2], d[3])
    np.testing.assert_array_equal(outs[0].cpu().numpy(), want1)
    np.testing.assert_array_equal(outs[1].cpu().numpy(), oracle.decode(obs[:4], frames[:4], trans, init))
    np.testing.assert_array_equal(torbi_amd.decode(small[0], small[1], d[2], d[3]).cpu().numpy(),
                                  oracle.decode(obs[:4], frames[:4], trans, init))
    d[2].mul_(0.5)                                                        # new version of the matrix: a new preparation
    want3 = oracle.decode(obs, frames, (trans * np.float32(0.5)).astype(np.float32), init)
    np.testing.assert_array_equal(torbi_amd.decode(*d).cpu().numpy(), want3)
    np.testing.assert_array_equal(torbi_amd.decode(*d).cpu().numpy(), want3)
    # the C entry: too small a buffer is refused, NULL is the plain call
    lib = torbi_amd._lib.load()
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    out = torch.empty((B, T), dtype=torch.int32, device=dev)
    one = (torbi_amd._lib.Batch * 1)(torbi_amd._lib.Batch(d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), ws.data_ptr(),
                                                          ws.numel(), B, T))
    import ctypes
    filled = ctypes.c_int(7)
    assert lib.torbi_hip_viterbi_decode_batches_prepared(one, 1, d[2].data_ptr(), d[3].data_ptr(), S, 0, None, 0, None,
                                                         ws.data_ptr(), 1024, ctypes.byref(filled)) == -2
    assert filled.value == 0
    assert lib.torbi_hip_viterbi_decode_batches_prepared(one, 1, d[2].data_ptr(), d[3].data_ptr(), S, 0, None, 0, None,
                                                         None, 0, ctypes.byref(filled)) == 0
    assert filled.value == 0
    np.testing.assert_array_equal(out.cpu().numpy(), want3)


@all_paths
def test_decode_pipeline_equals_serial_decodes():
    """torbi_amd.DecodePipeline: consecutive batches on alternating streams, private scratch."""
    dev = torch.device('cuda:0')
    pipe = torbi_amd.DecodePipeline(dev, depth=2)
    S = 360
    trans = torch.tensor(synth.scores(2, (S, S), seed=3), device=dev)
    init = torch.tensor(synth.scores(3, (S,), seed=3), device=dev)
    batches, outs = [], []
    for k, (B, T) in enumerate([(64, 30), (40, 17), (96, 45), (33, 8), (64, 30), (128, 12)]):
        obs = torch.tensor(synth.scores(1, (B, T, S), seed=k), device=dev)
        frames = torch.tensor(np.clip(synth.lengths(B, 1, T, seed=k), 1, T), device=dev)
        batches.append((obs, frames))
        outs.append(pipe.decode(obs, frames, trans, init))
    pipe.synchronize()
    for (obs, frames), got in zip(batches, outs):
        want = oracle.decode(obs.cpu().numpy(), frames.cpu().numpy(), trans.cpu().numpy(),
                             init.cpu().numpy(), num_threads=oracle.max_threads())
        assert np.array_equal(got.cpu().numpy(), want)


def test_fused_epsilon_clamp_is_bit_identical_to_the_torch_ops():
    """reference torbi/core.py:193-197: exp_, += tiny, log_ -- fused into one pass here."""
    dev = torch.device('cuda:0')
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.empty(8_000_003, device=dev).uniform_(-100.0, 2.0, generator=gen)
    x[:12] = torch.tensor([0.0, -0.0, -float('inf'), -87.3, -87.4, -103.0, -104.0, -1e-7, -1e-38, 1.0,
                           -88.0, -16.0], device=dev)
    probs = torch.rand(1_000_000, device=dev, generator=gen)
    x[100:100 + probs.numel()] = torch.log(probs)
    want = x.clone()
    torch.exp_(want)
    want += torch.finfo(torch.float32).tiny
    torch.log_(want)
    got = viterbi.epsilon_clamp_(x.clone())
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_fused_log_and_epsilon_clamp_is_bit_identical_to_the_torch_ops():
    """reference torbi/core.py:189-197 on probability inputs: log, exp_, += tiny, log_ -- one pass here, out of place
    (the caller's probabilities are not touched), NaN for negative inputs like torch.log."""
    dev = torch.device('cuda:0')
    gen = torch.Generator(device=dev).manual_seed(1)
    p = torch.rand(8_000_003, device=dev, generator=gen)
    p[:14] = torch.tensor([0.0, 1.0, 1e-45, 1e-38, 1.1754944e-38, 1e-30, 0.5, 0.99999994, 2.0, 1e30, float('inf'),
                           3e-39, 1e-20, 0.25], device=dev)
    p[14] = -1.0
    p[100:1_000_100] = torch.softmax(torch.randn(1000, 1000, device=dev, generator=gen) * 8.0, dim=-1).reshape(-1)
    keep = p.clone()
    want = torch.log(p)
    torch.exp_(want)
    want += torch.finfo(torch.float32).tiny
    torch.log_(want)
    got = viterbi.log_epsilon_clamp(p)
    assert got is not None and got.data_ptr() != p.data_ptr() and torch.equal(p.view(torch.int32), keep.view(torch.int32))
    same = (got.view(torch.int32) == want.view(torch.int32)) | (torch.isnan(got) & torch.isnan(want))
    assert bool(same.all())
    assert viterbi.log_epsilon_clamp(p[1:]) is None and viterbi.log_epsilon_clamp(p.double()) is None     # torch ops instead
    # ... and through from_probabilities: device-resident probabilities decode like host ones
    obs = torch.softmax(torch.randn(3, 17, 40, generator=torch.Generator().manual_seed(2)) * 3.0, dim=-1)
    a = torbi_amd.from_probabilities(obs.clone(), gpu=0)
    b = torbi_amd.from_probabilities(obs.to(dev), gpu=0)
    assert torch.equal(a.cpu(), b.cpu())


@all_paths
def test_posterior_rows_match_oracle_bitwise():
    B, T, S = 5, 23, 300
    obs, trans, init = synth.problem(B, T, S, seed=77)
    frames = np.array([23, 1, 2, 22, 10], np.int32)
    _, post = oracle.decode(obs, frames, trans, init, return_posterior=True)
    dev = torch.device('cuda:0')
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    torbi_amd.decode(torch.tensor(obs, device=dev), torch.tensor(frames, device=dev),
                     torch.tensor(trans, device=dev), torch.tensor(init, device=dev), workspace=ws)
    got = viterbi.read_posterior(ws, torch.tensor(frames), B, T, S).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), post.view(np.uint32))


def test_posterior_and_statistics_follow_the_route_the_decode_took():
    """torbi_hip_read_posterior / torbi_hip_scan_stats look at the route RECORD the last decode left in the workspace,
    not at what the batch's own shape and flags would choose: a 5-item batch decoded inside a time-resident launch
    group (its own route would be the generic kernels, whose posterior rows live elsewhere) reads back the oracle's
    posterior rows bit for bit, and the group's scan statistics are found without naming a path."""
    dev = torch.device('cuda:0')
    S = 300
    _, trans, init = synth.problem(1, 1, S, seed=78)
    d_trans, d_init = torch.tensor(trans, device=dev), torch.tensor(init, device=dev)
    shapes = [(300, 11), (5, 23)]
    batches, spaces, wanted = [], [], []
    for k, (B, T) in enumerate(shapes):
        obs = synth.scores(synth.STREAM_OBSERVATION, (B, T, S), seed=200 + k)
        frames = np.clip(synth.lengths(B, 1, T, seed=k), 1, T).astype(np.int32)
        frames[0] = T
        batches.append((torch.tensor(obs, device=dev), torch.tensor(frames, device=dev)))
        spaces.append(torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev))
        wanted.append(oracle.decode(obs, frames, trans, init, return_posterior=True, num_threads=oracle.max_threads()))
    for path in ('resident', 'cluster', 'dense'):
        got = viterbi.decode_batches([b[0] for b in batches], [b[1] for b in batches], d_trans, d_init, workspaces=spaces,
                                     path=path)
        for k, (B, T) in enumerate(shapes):
            np.testing.assert_array_equal(got[k].cpu().numpy(), wanted[k][0])
            post = viterbi.read_posterior(spaces[k], batches[k][1], B, T, S).cpu().numpy()      # no path named
            assert np.array_equal(post.view(np.uint32), wanted[k][1].view(np.uint32)), (path, k)
        stats = viterbi.scan_stats(spaces[0], shapes[0][0], shapes[0][1], S).cpu()
        if path != 'dense':
            assert int(stats[64]) > 0 and int(stats[127]) == 0       # wave passes counted, no cluster gave up waiting


def test_one_seed_per_item_after_shallow_scans_gives_the_same_indices():
    """TORBI_HIP_FEW_SEEDS: once an earlier time-resident launch with a matrix has reported shallow scans, later launches
    keep ONE explicit candidate per item instead of three (a third of the seed gathers' traffic, same speed on flat rows).
    The flag must not show in the results: flat and peaked rows, whole tiles and clusters, against the oracle."""
    dev = torch.device('cuda:0')
    S, T = 360, 14
    obs, trans, init = synth.problem(300, T, S, seed=31)
    rng = np.random.default_rng(5)
    peaked = obs.copy()
    centre = rng.integers(0, S, size=(300, T, 1))
    peaked -= ((np.abs(np.arange(S)[None, None, :] - centre) / 6.0) ** 2).astype(np.float32)
    frames = np.clip(synth.lengths(300, 1, T, seed=2), 1, T).astype(np.int32)
    frames[0] = T
    d_trans, d_init = torch.tensor(trans, device=dev), torch.tensor(init, device=dev)
    for data in (obs, peaked):
        want = oracle.decode(data, frames, trans, init, num_threads=oracle.max_threads())
        d_obs, d_frames = torch.tensor(data, device=dev), torch.tensor(frames, device=dev)
        for path in ('resident', 'cluster'):
            # unknown depth: the library's default (three seeds with whole tiles, one in clusters); shallow: one; deep: three
            for blocks, flag in ((None, 0), (1.0, 512), (float(S), 1024)):
                viterbi._depth_record(d_trans, S)[0] = blocks
                assert viterbi._seed_flag(d_trans, S) == flag
                got = torbi_amd.decode(d_obs, d_frames, d_trans, d_init, path=path)
                np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f'{path} seeds flag {flag}')
    torbi_amd.reset_path_state()


def test_timing_scope_covers_the_decode_on_the_device():
    """torbi_amd.timer (the stand-in for torchutil.time as torbi/core.py:200 uses it): from_probabilities accumulates
    under 'torbi'; with `device=` the scope is bracketed by HIP events and reports device time."""
    dev = torch.device('cuda:0')
    probs = torch.rand(40, 30, 96, generator=torch.Generator().manual_seed(0)).softmax(-1)
    trans = torch.rand(96, 96, generator=torch.Generator().manual_seed(1)).softmax(-1)
    torbi_amd.timer.reset()
    torbi_amd.from_probabilities(probs.clone(), transition=trans, gpu=0)
    assert torbi_amd.timer.results()['torbi'] > 0.0
    obs = torch.log(probs).to(dev)
    frames = torch.full((40,), 30, dtype=torch.int32, device=dev)
    with torbi_amd.timer.context('device side', device=dev):
        torbi_amd.decode(obs, frames, torch.log(trans).to(dev), torch.full((96,), -4.5, device=dev))
    got = torbi_amd.timer.results()
    assert 0.0 < got['device side'] < 5.0
    torbi_amd.timer.reset()


def test_fill_synthetic_matches_numpy_definition():
    for stream, seed, n, start in [(1, 0, 100003, 0), (2, 5, 4099, 17), (3, 1, 7, 1 << 33)]:
        got = viterbi.fill_synthetic((n,), stream, seed=seed, start=start).cpu().numpy()
        want = synth.scores(stream, (n,), seed=seed, start=start)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def path_score(obs, trans, init, idx, frames):
    """Score of each decoded path with the recurrence's own fp32 operation order."""
    B, T, S = obs.shape
    ar = torch.arange(B, device=obs.device)
    score = obs[ar, 0, idx[:, 0].long()] + init[idx[:, 0].long()]
    for t in range(1, T):
        live = t < frames
        prev, cur = idx[:, t - 1].long(), idx[:, t].long()
        new = obs[ar, t, cur] + (score + trans[cur, prev])
        score = torch.where(live, new, score)
    return score


@paths('auto', 'dense', 'resident')
def test_headline_shape_properties():
    """B=512, T=500, S=1440 (BASELINE config 3): too large for the oracle, so
    (1) the first 4 items equal the committed reference output (items are independent),
    (2) every decoded path's score, re-accumulated in the recurrence's own order, equals the
        maximum of that item's final posterior row bit for bit (the path is optimal),
    (3) permuting the batch permutes the result."""
    dev = torch.device('cuda:0')
    B, T, S = 512, 500, 1440
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
    trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
    init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    idx = torbi_amd.decode(obs, frames, trans, init, workspace=ws)
    post = viterbi.read_posterior(ws, frames, B, T, S)

    g = np.load(__import__('conftest').GOLDEN + '/golden_large.npz')
    assert np.array_equal(idx[:4].cpu().numpy(), g['g4_c3_first4_4x500x1440/indices'])

    assert int(idx.min()) >= 0 and int(idx.max()) < S
    score = path_score(obs, trans, init, idx, frames)
    assert torch.equal(score, post.max(dim=1).values)
    assert torch.equal(idx[:, -1].long(), post.argmax(dim=1))

    perm = torch.randperm(B, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    sub = perm[:96]
    idx2 = torbi_amd.decode(obs[sub].contiguous(), frames[sub].contiguous(), trans, init)
    assert torch.equal(idx2, idx[sub])


@paths('auto', 'dense', 'resident')
def test_large_state_shape_properties():
    """B=128, T=2000, S=4096 (BASELINE config 5, the large-S stress): the first 2 items equal the
    committed reference output; every decoded path re-scores to its item's final posterior
    maximum bit for bit; ragged lengths keep their tail fill."""
    dev = torch.device('cuda:0')
    B, T, S = 128, 2000, 4096
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
    trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
    init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    frames[5], frames[77] = 1234, 1
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    idx = torbi_amd.decode(obs, frames, trans, init, workspace=ws)
    post = viterbi.read_posterior(ws, frames, B, T, S)
    g = np.load(__import__('conftest').GOLDEN + '/golden_large.npz')
    assert np.array_equal(idx[:2].cpu().numpy(), g['g5_c5_first2_2x2000x4096/indices'])
    assert int(idx.min()) >= 0 and int(idx.max()) < S
    score = path_score(obs, trans, init, idx, frames)
    assert torch.equal(score, post.max(dim=1).values)
    last = idx[torch.arange(B, device=dev), (frames - 1).long()].long()
    assert torch.equal(last, post.argmax(dim=1))
    assert bool((idx[5, 1233:] == idx[5, 1233]).all()) and bool((idx[77] == idx[77, 0]).all())


@all_paths
def test_dispatcher_registration_matches_reference_call_site():
    """reference torbi/viterbi.py:53: torch.ops.torbi.viterbi_decode(observation, batch_frames,
    transition, initial) -- the same call reaches the HIP decode after torch_op.register()."""
    from torbi_amd import torch_op
    op = torch_op.register()
    assert torch_op.register() is not None            # idempotent
    dev = torch.device('cuda:0')
    obs, trans, init = synth.problem(40, 12, 96, seed=21)
    frames = np.clip(synth.lengths(40, 1, 12, seed=2), 1, 12)
    got = op(torch.tensor(obs, device=dev), torch.tensor(frames, device=dev),
             torch.tensor(trans, device=dev), torch.tensor(init, device=dev))
    assert got.dtype == torch.int32 and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), oracle.decode(obs, frames, trans, init))
    with pytest.raises(RuntimeError):                 # int64 lengths are rejected, as upstream
        op(torch.tensor(obs, device=dev), torch.tensor(frames, device=dev).long(),
           torch.tensor(trans, device=dev), torch.tensor(init, device=dev))


@all_paths
def test_ragged_batch_equals_oracle_decodes_of_every_file():
    """collate-style padded batch (reference collate.py:24-33) == the ORACLE's decode of each sequence alone
    (core.py:449-457 keeps the first `frames` indices of every row)."""
    S = 360
    lens = [37, 1, 120, 64, 2, 99]
    obs_full, trans, init = synth.problem(len(lens), max(lens), S, seed=11)
    for b, n in enumerate(lens):
        obs_full[b, n:] = 0.0          # zero padding as collate does
    got = gpu_decode(obs_full, lens, trans, init)
    for b, n in enumerate(lens):
        single = oracle.decode(obs_full[b:b + 1, :n], [n], trans, init)
        assert np.array_equal(got[b, :n], single[0])
        assert (got[b, n - 1:] == got[b, n - 1]).all()


@all_paths
def test_reference_toy_tests_on_gpu_and_host_tensors():
    """reference tests/test_core.py:7-46, both forms."""
    observation = torch.tensor([[0.25, 0.5, 0.25], [0.25, 0.25, 0.5], [0.33, 0.33, 0.33]]).unsqueeze(dim=0)
    transition = torch.tensor([[0.5, 0.25, 0.25], [0.33, 0.34, 0.33], [0.25, 0.25, 0.5]])
    initial = torch.tensor([0.4, 0.35, 0.25])
    bins = torbi_amd.from_probabilities(
        observation=observation, transition=transition, initial=initial, log_probs=False)
    assert bins.device.type == 'cpu' and bins.dtype == torch.int32
    assert (bins == torch.tensor([1, 2, 2])).all()
    bins = torbi_amd.from_probabilities(
        observation=observation.to('cuda:0'), transition=transition.to('cuda:0'),
        initial=initial.to('cuda:0'), log_probs=False, gpu=0)
    assert (bins == torch.tensor([1, 2, 2]).to('cuda:0')).all()
    assert torbi_amd.from_probabilities(observation).tolist() == [[1, 2, 0]]
    assert torbi_amd.from_probabilities(
        observation, batch_frames=torch.tensor([2]), transition=transition,
        initial=initial).tolist() == [[1, 2, 2]]


@all_paths
def test_from_probabilities_equals_decode_of_same_device_preprocessing():
    """pin the defaults and the epsilon round trip with this device's ops."""
    import math
    B, T, S = 3, 20, 50
    probs = torch.rand(B, T, S, generator=torch.Generator().manual_seed(1)).softmax(-1)
    got = torbi_amd.from_probabilities(probs.clone(), gpu=0)
    tiny = torch.finfo(torch.float32).tiny
    x = torch.log(probs).to('cuda:0')
    x = torch.log(torch.exp(x) + tiny)
    init = torch.full((S,), math.log(1. / S + tiny), device='cuda:0')
    trans = torch.full((S, S), math.log(1. / S), device='cuda:0')
    frames = torch.full((B,), T, dtype=torch.int32, device='cuda:0')
    want = torbi_amd.decode(x, frames, trans, init)
    assert torch.equal(got, want)
    ref = oracle.decode(x.cpu().numpy(), frames.cpu().numpy(), trans.cpu().numpy(), init.cpu().numpy())
    assert np.array_equal(want.cpu().numpy(), ref)


def _oracle_for_file(observation, transition_probs, states):
    """What from_files_to_files(log_probs=True) feeds the operator for one file, decoded by the oracle: the
    observation goes through THIS device's epsilon round trip, the transition through the
    host's log(p + tiny) (reference core.py:341-347), the initial defaults to log(1/S + tiny) (core.py:161-166)."""
    import math
    tiny = torch.finfo(torch.float32).tiny
    x = observation.to('cuda:0', dtype=torch.float32)
    x = torch.log(torch.exp(x) + tiny).cpu().numpy()[None]
    trans = torch.log(transition_probs + tiny).numpy()
    init = np.full((states,), math.log(1. / states + tiny), dtype=np.float32)
    return oracle.decode(x, [x.shape[1]], trans, init, num_threads=min(oracle.max_threads(), max(1, states // 16)))[0]


@all_paths
def test_files_round_trip(tmp_path):
    """from_files_to_files / from_file_to_file write, per file, the oracle's decode of that file alone
    (reference core.py:310-368, 211-307)."""
    S = 40
    ins, outs = [], []
    gen = torch.Generator().manual_seed(3)
    for k, n in enumerate([5, 17, 1, 9]):
        f = tmp_path / f'in{k}.pt'
        torch.save(torch.rand(n, S, generator=gen).log_softmax(-1), f)
        ins.append(f)
        outs.append(tmp_path / f'out{k}.pt')
    tf = tmp_path / 'transition.pt'
    torch.save(torch.rand(S, S, generator=gen).softmax(-1), tf)
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0)
    for fin, fout in zip(ins, outs):
        got = torch.load(fout)
        n = torch.load(fin).shape[0]
        assert got.shape == (n,) and got.dtype == torch.int32
        assert np.array_equal(got.numpy(), _oracle_for_file(torch.load(fin), torch.load(tf), S))
    # length-bucketed batching writes the same files
    outs2 = [tmp_path / f'sorted{k}.pt' for k in range(len(ins))]
    saved = torbi_amd.core.BATCH_SIZE
    torbi_amd.core.BATCH_SIZE = 2
    try:
        torbi_amd.from_files_to_files(ins, outs2, transition_file=tf, log_probs=True, gpu=0,
                                      lengths=[5, 17, 1, 9])
    finally:
        torbi_amd.core.BATCH_SIZE = saved
    for a, b2 in zip(outs, outs2):
        assert torch.equal(torch.load(a), torch.load(b2))
    torbi_amd.from_file_to_file(ins[1], tmp_path / 'single.pt', log_probs=True, gpu=0)
    assert torch.load(tmp_path / 'single.pt').shape == (1, 17)


# ---- several batches per call (torbi_hip_viterbi_decode_batches) and the time-resident path -----------------

def _device_problem(B, T, S, seed, dev, ragged=True):
    obs, trans, init = synth.problem(B, T, S, seed=seed)
    frames = np.clip(synth.lengths(B, 1, T, seed=seed + 1), 1, T) if ragged else np.full(B, T, np.int32)
    if ragged:
        frames[0] = T
    return obs, frames.astype(np.int32), trans, init


@all_paths
@pytest.mark.parametrize('S', [64, 130, 360, 1440, 1442, 2048, 2052, 4096])
def test_decode_batches_equals_oracle_per_batch(S):
    """A group of batches with different sizes and lengths (a many-file job, reference torbi/core.py:417-457)
    through ONE call: every batch equals the oracle decode of that batch alone, whatever path the group takes
    (forced 'resident': one forward launch for the whole group; otherwise batch after batch)."""
    dev = torch.device('cuda:0')
    shapes = [(40, 9), (17, 23), (1, 5), (96, 4), (33, 1), (16, 12)] if S < 1000 else [(40, 7), (17, 9), (3, 4)]
    _, trans, init = synth.problem(1, 1, S, seed=S)
    obs_list, frame_list, want = [], [], []
    for k, (B, T) in enumerate(shapes):
        obs, frames, _, _ = _device_problem(B, T, S, seed=100 * k + S, dev=dev)
        obs_list.append(torch.as_tensor(obs).to(dev))
        frame_list.append(torch.as_tensor(frames).to(dev))
        want.append(oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads()))
    got = viterbi.decode_batches(obs_list, frame_list, torch.as_tensor(trans).to(dev), torch.as_tensor(init).to(dev))
    for k in range(len(shapes)):
        np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=f'batch {k} {shapes[k]}')


def test_decode_batches_auto_goes_resident_when_the_group_fills_the_chip():
    """AUTO counts the 16-item tiles of the whole group: 8 batches of 272 items are 136 tiles >= half the compute
    units -> one time-resident launch (phase record: route 3, 8 batches); two of them are decoded one by one."""
    dev = torch.device('cuda:0')
    S, T, B = 360, 12, 272
    _, trans, init = synth.problem(1, 1, S, seed=5)
    obs_list, frame_list, want = [], [], []
    for k in range(8):
        obs, frames, _, _ = _device_problem(B, T, S, seed=40 + k, dev=dev)
        obs_list.append(torch.as_tensor(obs).to(dev))
        frame_list.append(torch.as_tensor(frames).to(dev))
        want.append(oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads()))
    d_trans, d_init = torch.as_tensor(trans).to(dev), torch.as_tensor(init).to(dev)
    prof = []
    got = viterbi.decode_batches(obs_list, frame_list, d_trans, d_init, path='auto', _profile=prof)
    assert int(prof[3]) == 3 and int(prof[5]) == 8 and int(prof[2]) == 1
    for k in range(8):
        np.testing.assert_array_equal(got[k].cpu().numpy(), want[k])
    # a group that fills less than half the chip: still ONE launch, tiles split over clusters of workgroups
    got = viterbi.decode_batches(obs_list[:2], frame_list[:2], d_trans, d_init, path='auto', _profile=prof)
    assert int(prof[3]) == 5 and int(prof[5]) == 2 and int(prof[2]) == 1
    for k in range(2):
        np.testing.assert_array_equal(got[k].cpu().numpy(), want[k])
    # a narrow band goes to the band kernel (csrc/band_forward.hpp): one batch, a few, a group that fills the chip -- one launch
    band = synth.banded_transition(S, 12.0)
    d_band = torch.as_tensor(band).to(dev)
    got = viterbi.decode_batches(obs_list, frame_list, d_band, d_init, path='auto', _profile=prof)
    assert int(prof[3]) == 8 and int(prof[5]) == 8 and int(prof[2]) == 1
    for k in (0, 7):
        ref = oracle.decode(obs_list[k].cpu().numpy(), frame_list[k].cpu().numpy(), band, init,
                            num_threads=oracle.max_threads())
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref)
    viterbi.decode_batches(obs_list[:1], frame_list[:1], d_band, d_init, path='auto', _profile=prof)
    assert int(prof[3]) == 8
    got = viterbi.decode_batches(obs_list[:3], frame_list[:3], d_band, d_init, path='auto', _profile=prof)
    assert int(prof[3]) == 8 and int(prof[5]) == 3
    ref = oracle.decode(obs_list[2].cpu().numpy(), frame_list[2].cpu().numpy(), band, init, num_threads=oracle.max_threads())
    np.testing.assert_array_equal(got[2].cpu().numpy(), ref)


@paths('auto', 'dense', 'resident')
@pytest.mark.parametrize('B', [40, 270])
@pytest.mark.parametrize('S', [360, 1440, 4096])
@pytest.mark.parametrize('kind', ['some', 'rows', 'all'])
def test_minus_inf_observations_on_the_large_batch_paths(B, S, kind):
    """-inf observation entries, whole -inf observation rows and all -inf observations at batch sizes that take
    the value-only paths (the pruned bound sees thr = -inf, keys of -inf outputs, tn + thr > best with -inf)."""
    T = 6 if S < 4096 else 4
    obs, trans, init = synth.problem(B, T, S, seed=B + S)
    rng = np.random.default_rng(B * S)
    if kind == 'some':
        obs = np.where(rng.random(obs.shape) < 0.3, -np.inf, obs).astype(np.float32)
        init = np.where(rng.random(S) < 0.5, -np.inf, init).astype(np.float32)
    elif kind == 'rows':
        dead = rng.random((B, T)) < 0.4
        dead[0, 1] = True
        obs = np.where(dead[:, :, None], -np.inf, obs).astype(np.float32)
    else:
        obs = np.full_like(obs, -np.inf)
    frames = np.clip(synth.lengths(B, 1, T, seed=S), 1, T)
    frames[0] = T
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    np.testing.assert_array_equal(gpu_decode(obs, frames, trans, init), want)


@all_paths
def test_headline_batch_against_the_oracle_on_random_items(forward):
    """B=512, T=500, S=1440 (BASELINE config 3): 64 randomly chosen items of the full batch against the oracle
    (lowest-index ties included: the path-score property of test_headline_shape_properties accepts any optimal
    path).  Host cost: ~1 s per item with all threads on the GPU box."""
    if forward == 'dense':
        pytest.skip('the dense kernel sees the same items in test_headline_shape_properties; 40 s per decode')
    dev = torch.device('cuda:0')
    B, T, S = 512, 500, 1440
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, device=dev)
    trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, device=dev)
    init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    idx = torbi_amd.decode(obs, frames, trans, init)
    pick = np.sort(np.random.default_rng(20).choice(B, size=64, replace=False))
    if 'want' not in _HEADLINE_ORACLE:      # the same items under every forward path: decode them on the host once
        _HEADLINE_ORACLE['want'] = oracle.decode(
            obs[torch.as_tensor(pick).to(dev)].cpu().numpy(), np.full(64, T, np.int32), trans.cpu().numpy(),
            init.cpu().numpy(), num_threads=oracle.max_threads(), mode=1)
    np.testing.assert_array_equal(idx.cpu().numpy()[pick], _HEADLINE_ORACLE['want'])


_HEADLINE_ORACLE = {}


@paths('auto', 'dense', 'resident')
def test_inference_mode_is_supported():
    """Tensors created under torch.inference_mode() have no version counter; decode / from_probabilities must
    not depend on one (the reference works there)."""
    dev = torch.device('cuda:0')
    B, T, S = 40, 9, 96
    obs, trans, init = synth.problem(B, T, S, seed=3)
    frames = np.clip(synth.lengths(B, 1, T, seed=4), 1, T)
    want = oracle.decode(obs, frames, trans, init)
    with torch.inference_mode():
        d = [torch.as_tensor(x).to(dev) * 1 for x in (obs, trans, init)]
        f = torch.as_tensor(frames).to(dev)
        ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
        for _ in range(2):
            got = torbi_amd.decode(d[0], f, d[1], d[2], workspace=ws, reuse_preparation=True)
            np.testing.assert_array_equal(got.cpu().numpy(), want)
        probs = torch.rand(2, 7, 30, generator=torch.Generator().manual_seed(0)).softmax(-1)
        trans_p = torch.rand(30, 30, generator=torch.Generator().manual_seed(1)).softmax(-1)
        inside = torbi_amd.from_probabilities(probs.clone(), transition=trans_p, gpu=0)
        assert torch.ops is not None and inside.shape == (2, 7)
        op = __import__('torbi_amd.torch_op', fromlist=['register']).register()
        np.testing.assert_array_equal(op(d[0], f, d[1], d[2]).cpu().numpy(), want)
    outside = torbi_amd.from_probabilities(probs.clone(), transition=trans_p, gpu=0)
    assert torch.equal(inside.cpu(), outside.cpu())


@all_paths
def test_concurrent_host_threads_on_separate_streams():
    """SURVEY 8(b) threading: two host threads, each with its own stream (and its own device when there are
    two), decode a banded and a dense matrix at the same time, each naming its forward path in the call.
    Nothing process-wide is involved, so neither sees the other's choice; both equal the oracle."""
    import threading
    n_dev = torch.cuda.device_count()
    S, T = 360, 16
    jobs = []
    for k, (kind, path, B) in enumerate([('banded', 'dense', 64), ('dense', 'cluster', 48)]):
        obs, frames, trans, init = _device_problem(B, T, S, seed=7 + k, dev=None)
        if kind == 'banded':
            trans = synth.banded_transition(S, 12.0)
        jobs.append(dict(obs=obs, frames=frames, trans=trans, init=init, path=path,
                         dev=torch.device('cuda', k % n_dev),
                         want=oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())))
    errors = []

    def work(job):
        try:
            dev = job['dev']
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream(device=dev)
            d = [torch.as_tensor(job[name]).to(dev) for name in ('obs', 'frames', 'trans', 'init')]
            ws = torch.empty(viterbi.workspace_bytes(*job['obs'].shape), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(stream):
                for _ in range(25):
                    got = torbi_amd.decode(*d, workspace=ws, reuse_preparation=True, path=job['path'])
                    if not np.array_equal(got.cpu().numpy(), job['want']):
                        errors.append(f"{job['path']}: indices differ from the oracle")
                        return
        except Exception as exc:      # surfaced below: an exception in a thread must fail the test
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(job,)) for job in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@all_paths
def test_host_threads_share_one_kept_preparation():
    """Four host threads, each on its own stream, call decode() WITHOUT a workspace on the SAME transition tensor at the same
    time: one of them fills the preparation kept with the tensor (torbi_amd/viterbi.py::_Preparation), the others wait for
    that call to be enqueued and order their streams behind it with its event; every result equals the oracle, and one
    buffer serves them all."""
    import threading
    from torbi_amd import state
    dev = torch.device('cuda:0')
    S, T = 360, 12
    _, trans, init = synth.problem(1, 1, S, seed=31)
    d_trans, d_init = torch.tensor(trans, device=dev), torch.tensor(init, device=dev)
    jobs = []
    for k, B in enumerate((40, 33, 64, 21)):
        obs = synth.scores(synth.STREAM_OBSERVATION, (B, T, S), seed=300 + k)
        frames = np.clip(synth.lengths(B, 1, T, seed=k), 1, T)
        jobs.append((obs, frames, oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())))
    if viterbi.forward_path(40, S) not in viterbi.TIME_RESIDENT:
        pytest.skip('this forced path keeps no preparation')
    errors, start = [], threading.Barrier(len(jobs))

    def work(job):
        try:
            obs, frames, want = job
            stream = torch.cuda.Stream(device=dev)
            d_obs, d_frames = torch.tensor(obs, device=dev), torch.tensor(frames, device=dev)
            torch.cuda.synchronize(dev)
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(10):
                    got = torbi_amd.decode(d_obs, d_frames, d_trans, d_init)
                    if not np.array_equal(got.cpu().numpy(), want):
                        errors.append('indices differ from the oracle')
                        return
        except Exception as exc:
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(job,)) for job in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    kept = [v for k, v in state.peek(d_trans).items() if isinstance(k, tuple) and k[0] == 'preparation']
    assert len(kept) == 1 and kept[0].filled is not None


# ---- vectors produced by the reference's own Python on its CPU operator (tests/golden/generate_api.py) -------

API = np.load(__import__('conftest').GOLDEN + '/golden_api.npz')


def _write_api_files(tmp_path, tag):
    count = int(API[f'files_{tag}/count'])
    ins, outs = [], []
    for k in range(count):
        f = tmp_path / f'in{k}.pt'
        torch.save(torch.as_tensor(API[f'files_{tag}/in{k}']), f)
        ins.append(f)
        outs.append(tmp_path / f'out{k}.pt')
    tf = tmp_path / 'transition.pt'
    torch.save(torch.as_tensor(API[f'files_{tag}/transition']), tf)
    return ins, outs, tf


@all_paths
def test_from_files_to_files_equals_the_reference_outputs(tmp_path, monkeypatch):
    """the files the reference's from_files_to_files wrote (real torbi Python, CPU operator, batch
    size 3: three batches) for seven ragged inputs -- same shapes, dtypes and indices here."""
    monkeypatch.setattr(torbi_amd.core, 'BATCH_SIZE', int(API['files_plain/batch_size']))
    ins, outs, tf = _write_api_files(tmp_path, 'plain')
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0)
    for k, f in enumerate(outs):
        got = torch.load(f)
        want = API[f'files_plain/out{k}']
        assert got.dtype == torch.int32 and tuple(got.shape) == want.shape
        np.testing.assert_array_equal(got.numpy(), want, err_msg=f'file {k}')


@all_paths
def test_chunked_from_files_to_files_equals_the_reference_outputs(tmp_path, monkeypatch):
    """chunked decoding (reference torbi/chunk.py with MIN_CHUNK_SIZE = 8): files are cut at
    the same frames, decoded as extra batch rows and joined; outputs equal the reference's."""
    monkeypatch.setattr(torbi_amd.core, 'BATCH_SIZE', int(API['files_chunk/batch_size']))
    monkeypatch.setattr(torbi_amd.core, 'MIN_CHUNK_SIZE', int(API['chunk/min_chunk_size']))
    ins, outs, tf = _write_api_files(tmp_path, 'chunk')
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0)
    for k, f in enumerate(outs):
        got = torch.load(f)
        want = API[f'files_chunk/out{k}']
        assert got.dtype == torch.int32 and tuple(got.shape) == want.shape
        np.testing.assert_array_equal(got.numpy(), want, err_msg=f'file {k}')


@all_paths
def test_from_probabilities_equals_the_reference_outputs():
    """reference from_probabilities (CPU) on probability and log-probability inputs, with given
    and with default transition / initial (the epsilon round trip runs on different devices: SURVEY 0.5 -- these
    inputs were checked to decode identically)."""
    obs = torch.as_tensor(API['probs/observation'])
    trans = torch.as_tensor(API['probs/transition'])
    init = torch.as_tensor(API['probs/initial'])
    frames = torch.as_tensor(API['probs/batch_frames'])
    got = torbi_amd.from_probabilities(obs.clone(), frames, trans, init, log_probs=False, gpu=0)
    np.testing.assert_array_equal(got.cpu().numpy(), API['probs/indices'])
    got = torbi_amd.from_probabilities(obs.clone(), gpu=0)
    np.testing.assert_array_equal(got.cpu().numpy(), API['probs/indices_defaults'])
    got = torbi_amd.from_probabilities(torch.log(obs), frames, torch.log(trans), torch.log(init), log_probs=True, gpu=0)
    np.testing.assert_array_equal(got.cpu().numpy(), API['probs/indices_log'])


def test_a_large_host_batch_of_probabilities_is_logged_like_upstream(monkeypatch):
    """core.py:189-191: upstream takes log() of the observation where it lives, BEFORE the device move -- a host batch is
    logged by the host.  Large float32 host batches take that log into a pooled pinned buffer (core._host_log): the same
    torch kernel, so the decode equals the one of `torch.log(x)` handed over as log-probabilities; the caller's tensor is
    left alone; the buffer goes back to the pool and is reused by the next call; a small batch and a float64 batch go the
    plain way and give the same indices."""
    from torbi_amd import core, slabs
    monkeypatch.setattr(core, 'HOST_LOG_POOL_BYTES', 1 << 16)
    B, T, S = 24, 30, 360
    gen = torch.Generator().manual_seed(5)
    probs = torch.rand(B, T, S, generator=gen).mul_(5.0).softmax(-1)
    probs[:, :, 7] = 0.0
    trans = torch.rand(S, S, generator=gen).mul_(4.0).softmax(-1)
    frames = torch.as_tensor(synth.lengths(B, 1, T).astype(np.int32))
    keep = probs.clone()
    want = torbi_amd.from_probabilities(torch.log(probs), frames, torch.log(trans), log_probs=True, gpu=0).cpu()
    held = slabs.pool(None).held_bytes()
    got = torbi_amd.from_probabilities(probs, frames, trans, gpu=0).cpu()
    assert torch.equal(got, want) and torch.equal(probs, keep)
    torch.cuda.synchronize()
    grown = slabs.pool(None).held_bytes()
    assert grown >= held + probs.numel() * 4 or held >= probs.numel() * 4          # the pooled buffer came back
    again = torbi_amd.from_probabilities(probs, frames, trans, gpu=0).cpu()
    torch.cuda.synchronize()
    assert torch.equal(again, want) and slabs.pool(None).held_bytes() == grown    # ... and was used again
    monkeypatch.setattr(core, 'HOST_LOG_POOL_BYTES', 1 << 40)
    assert torch.equal(torbi_amd.from_probabilities(probs, frames, trans, gpu=0).cpu(), want)
    assert torch.equal(torbi_amd.from_probabilities(probs[:, ::2], frames.clamp(max=15), trans, gpu=0).cpu(),
                       torbi_amd.from_probabilities(torch.log(probs[:, ::2]), frames.clamp(max=15), torch.log(trans),
                                                    log_probs=True, gpu=0).cpu())


# ---- BASELINE configs[3] scaled down: a ragged many-file job through from_files_to_files ---------------------

def _ragged_job(tmp_path, count, S, seed, shortest=100, longest=900):
    lengths = synth.lengths(count, shortest, longest, seed=seed).tolist()
    gen = torch.Generator().manual_seed(seed)
    block = torch.rand(1000, S, generator=gen).mul_(6.0).log_softmax(-1)
    ins, outs = [], []
    for k, n in enumerate(lengths):
        f = tmp_path / f'in{k}.pt'
        start = (37 * k) % 100
        torch.save((block[start:start + n] + 0.01 * (k % 7)).log_softmax(-1).clone(), f)
        ins.append(f)
        outs.append(tmp_path / f'out{k}.pt')
    tf = tmp_path / 'transition.pt'
    torch.save(torch.rand(S, S, generator=gen).mul_(4.0).softmax(-1), tf)
    return lengths, ins, outs, tf


def test_many_file_job_every_file_equals_the_oracle(tmp_path, monkeypatch):
    """BASELINE configs[3] scaled down fivefold in time: 2100 sequences of 20..180 frames over 256 states, batches of 512
    in file order (five batches -> one launch group) and again length-bucketed: EVERY output file equals the oracle's
    decode of that file alone."""
    S, count = 256, 2100
    lengths, ins, outs, tf = _ragged_job(tmp_path, count, S, seed=4, shortest=20, longest=180)
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0)
    trans = torch.load(tf)
    # the oracle decodes the files 100 at a time as one ragged batch (its batch loop is serial, viterbi.cpp:65: the same
    # per-file decodes, without 2 100 thread-team start-ups); inputs through THIS device's epsilon round trip, the
    # transition through the host's log(p + tiny), the default initial (_oracle_for_file)
    import math
    tiny = torch.finfo(torch.float32).tiny
    log_trans = torch.log(trans + tiny).numpy()
    init = np.full((S,), math.log(1. / S + tiny), dtype=np.float32)
    for first in range(0, count, 100):
        group = range(first, min(first + 100, count))
        longest = max(lengths[k] for k in group)
        padded = np.zeros((len(group), longest, S), np.float32)
        for row, k in enumerate(group):
            x = torch.load(ins[k]).to('cuda:0', dtype=torch.float32)
            padded[row, :lengths[k]] = torch.log(torch.exp(x) + tiny).cpu().numpy()
        want = oracle.decode(padded, [lengths[k] for k in group], log_trans, init, num_threads=min(oracle.max_threads(), S // 16))
        for row, k in enumerate(group):
            got = torch.load(outs[k])
            assert got.dtype == torch.int32 and got.shape == (lengths[k],)
            assert np.array_equal(got.numpy(), want[row, :lengths[k]]), f'file {k} ({lengths[k]} frames)'
    outs2 = [tmp_path / f'bucketed{k}.pt' for k in range(count)]
    torbi_amd.from_files_to_files(ins, outs2, transition_file=tf, log_probs=True, gpu=0, lengths=lengths)
    for a, b in zip(outs, outs2):
        assert torch.equal(torch.load(a), torch.load(b))
    # the host link's ring of pinned chunks (core.py::_Staging.upload_rows): chunks of a few rows (a 512-file batch in ~90
    # pieces through three chunks), chunks smaller than ONE row (a chunk per row), and whole-batch pinned slabs instead
    for chunk_bytes, chunks in ((1 << 20, 3), (1 << 12, 2), (1 << 28, 0)):
        monkeypatch.setattr(torbi_amd.core, 'RING_CHUNK_BYTES', chunk_bytes)
        monkeypatch.setattr(torbi_amd.core, 'RING_CHUNKS', chunks)
        outs3 = [tmp_path / f'ring{k}.pt' for k in range(count)]
        torbi_amd.from_files_to_files(ins, outs3, transition_file=tf, log_probs=True, gpu=0)
        for a, b in zip(outs, outs3):
            assert torch.equal(torch.load(a), torch.load(b)), (chunk_bytes, chunks)


def test_many_file_job_on_probability_files(tmp_path, monkeypatch):
    """The reference's DEFAULT call -- files of probabilities, log_probs=False (core.py:310-318) -- takes the same staged
    route as log-probability files: rows through the ring of pinned chunks into a pooled device slab, log() and the epsilon
    round trip (core.py:189-197) as ONE pass in place there.  Every output equals the reference loader's (torch.load +
    collate + from_probabilities' own log / clamp) and whole-batch staging's; every 9th file equals the oracle's decode of that
    file alone; the default uniform model (no transition file) likewise."""
    import math
    S, count = 256, 700
    lengths = synth.lengths(count, 20, 120, seed=11).tolist()
    gen = torch.Generator().manual_seed(11)
    block = torch.rand(300, S, generator=gen).mul_(6.0).softmax(-1)
    block[:, 5] = 0.0                                       # (probability zero: log -> -inf -> log(tiny) behind the round trip)
    ins = []
    for k, n in enumerate(lengths):
        ins.append(tmp_path / f'in{k}.pt')
        torch.save(block[(13 * k) % 150:(13 * k) % 150 + n].roll(k, dims=1).clone(), ins[-1])
    tf = tmp_path / 'transition.pt'
    trans = torch.rand(S, S, generator=gen).mul_(4.0).softmax(-1)
    trans[trans < 0.001] = 0.0                              # (-inf entries for the operator)
    torch.save(trans, tf)
    tiny = torch.finfo(torch.float32).tiny
    init = np.full((S,), math.log(1. / S + tiny), dtype=np.float32)
    for model in (dict(transition_file=tf), dict()):
        outs = [tmp_path / f'out{k}.pt' for k in range(count)]
        torbi_amd.from_files_to_files(ins, outs, log_probs=False, gpu=0, **model)
        want = [torch.load(f) for f in outs]
        log_trans = torch.log(trans).numpy() if model else np.full((S, S), np.float32(math.log(1. / S)), dtype=np.float32)
        for k in range(0, count, 9):
            x = torch.log(torch.load(ins[k]).to('cuda:0'))
            x = torch.log(torch.exp(x) + tiny).cpu().numpy()[None]
            assert np.array_equal(want[k].numpy(), oracle.decode(x, [lengths[k]], log_trans, init)[0]), (bool(model), k)
        with monkeypatch.context() as patch:
            patch.setattr(torbi_amd.core, 'RING_CHUNKS', 0)
            torbi_amd.from_files_to_files(ins, outs, log_probs=False, gpu=0, lengths=lengths, **model)
            assert all(torch.equal(torch.load(f), w) for f, w in zip(outs, want)), bool(model)
        with monkeypatch.context() as patch:
            patch.setattr(torbi_amd.core, 'DIRECT_FILE_IO', False)
            torbi_amd.from_files_to_files(ins, outs, log_probs=False, gpu=0, **model)
            assert all(torch.equal(torch.load(f), w) for f, w in zip(outs, want)), bool(model)


@pytest.mark.parametrize('count', [600, pytest.param(2100, marks=pytest.mark.slow)])
def test_many_file_job_through_the_reference_loader(tmp_path, monkeypatch, count):
    """The same ragged job with the reference's host path (torch.load + collate in a DataLoader, saves on the calling
    thread) instead of the direct file reader (torbi_amd/fastio.py) and the saver threads: identical output files.
    600 files (two batches) in the default run; the 2 100-file job is the slow instance (the loader delivers 1.5 GB/s)."""
    S = 256
    lengths, ins, outs, tf = _ragged_job(tmp_path, count, S, seed=4, shortest=20, longest=180)
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0)
    monkeypatch.setattr(torbi_amd.core, 'DIRECT_FILE_IO', False)
    monkeypatch.setattr(torbi_amd.core, 'SAVE_THREADS', 0)
    outs3 = [tmp_path / f'loader{k}.pt' for k in range(count)]
    torbi_amd.from_files_to_files(ins, outs3, transition_file=tf, log_probs=True, gpu=0, num_workers=2)
    trans = torch.load(tf)
    for k, (a, b) in enumerate(zip(outs, outs3)):
        got = torch.load(b)
        assert torch.equal(torch.load(a), got)
        if k % 10 == 0:
            assert np.array_equal(got.numpy(), _oracle_for_file(torch.load(ins[k]), trans, S)), f'file {k}'


def test_many_file_job_with_the_default_uniform_transition(tmp_path):
    """from_files_to_files(log_probs=True, transition_file=None, gpu=0): the reference's default model (core.py:175-180)
    takes the uniform-transition entry, which decodes on the staging side's preparation stream -- the consumer's copy of
    the indices has to be ordered behind it (round-3 review: the last batch of a job, drained right behind its
    host-to-device copy, was saved unwritten).  1 100 ragged files = 3 batches of 512; every output equals the oracle's
    decode of that file with the materialised matrix."""
    import math
    S, count = 256, 1100
    lengths, ins, outs, _ = _ragged_job(tmp_path, count, S, seed=11, shortest=20, longest=180)
    tiny = torch.finfo(torch.float32).tiny
    trans = np.full((S, S), np.float32(math.log(1. / S)), dtype=np.float32)
    init = np.full((S,), math.log(1. / S + tiny), dtype=np.float32)
    want = []
    for k in range(count):
        x = torch.log(torch.exp(torch.load(ins[k]).to('cuda:0')) + tiny).cpu().numpy()[None]
        want.append(oracle.decode(x, [x.shape[1]], trans, init, num_threads=8)[0])
    for rep in range(2):              # (the second job finds the pooled slabs and the kept pipeline of the first)
        torbi_amd.from_files_to_files(ins, outs, log_probs=True, gpu=0)
        for k in range(count):
            got = torch.load(outs[k])
            assert got.dtype == torch.int32 and got.shape == (lengths[k],)
            assert np.array_equal(got.numpy(), want[k]), f'file {k} ({lengths[k]} frames), job {rep}'
        for f in outs:
            f.unlink()


def test_many_file_job_frees_its_scratch_unless_asked_to_keep_it(tmp_path, monkeypatch):
    """Round-3 advisor: the launch-group pipeline (2 x GROUP_SIZE workspaces) and the staging slabs used to stay allocated
    for the life of the process.  By default a job now frees them when it ends; KEEP_JOB_MEMORY keeps them for the next
    job and release_job_memory() drops them."""
    from torbi_amd import core, slabs
    dev = torch.device('cuda', 0)
    S = 256
    lengths, ins, outs, tf = _ragged_job(tmp_path, 700, S, seed=5, shortest=20, longest=120)
    core.release_job_memory()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    floor = torch.cuda.memory_allocated(dev)
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0)
    assert not core._job_pipelines and slabs.pool(dev).held_bytes() == 0 and slabs.pool(None).held_bytes() == 0
    assert torch.cuda.memory_allocated(dev) - floor < 64 << 20, 'a finished job keeps device memory'
    first = [torch.load(f) for f in outs]
    monkeypatch.setattr(core, 'KEEP_JOB_MEMORY', True)
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0)
    assert list(core._job_pipelines) == [str(dev)] and slabs.pool(dev).held_bytes() > 0
    assert all(torch.equal(a, torch.load(f)) for a, f in zip(first, outs))
    core.release_job_memory()
    assert not core._job_pipelines and slabs.pool(dev).held_bytes() == 0


def test_many_file_job_at_1440_states(tmp_path):
    """BASELINE configs[3] at its own state count, 560 sequences (two batches: 512 + 48): in-order and
    length-bucketed batching write identical files, and 40 randomly chosen files equal the oracle's decode of
    that file alone (about 1 s of host time each)."""
    S, count = 1440, 560
    lengths, ins, outs, tf = _ragged_job(tmp_path, count, S, seed=9)
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0)
    outs2 = [tmp_path / f'bucketed{k}.pt' for k in range(count)]
    torbi_amd.from_files_to_files(ins, outs2, transition_file=tf, log_probs=True, gpu=0, lengths=lengths)
    trans = torch.load(tf)
    for k in range(count):
        a, b = torch.load(outs[k]), torch.load(outs2[k])
        assert a.dtype == torch.int32 and a.shape == (lengths[k],) and torch.equal(a, b)
    for k in np.random.default_rng(3).choice(count, size=40, replace=False):
        want = _oracle_for_file(torch.load(ins[k]), trans, S)
        assert np.array_equal(torch.load(outs[k]).numpy(), want), f'file {k} ({lengths[k]} frames)'


def test_many_file_job_with_the_pitch_transition_file(tmp_path):
    """The reference's evaluation as it calls the library (torbi/evaluate/core.py:97-103): a file of pitch transition
    PROBABILITIES and log_probs=True, i.e. the operator decodes with log(p + tiny) -- log(tiny) outside the band
    (torbi/core.py:341-347).  2 100 ragged sequences at 1440 states: the launch groups run the whole-tile band kernel with a
    constant outside the band, every output has its file's length, and 30 files equal the oracle's decode of that file alone."""
    S, count = 1440, 2100
    lengths = synth.lengths(count, 20, 60, seed=12).tolist()
    gen = torch.Generator().manual_seed(12)
    block = torch.rand(200, S, generator=gen).mul_(6.0)
    ins, outs = [], []
    for k, n in enumerate(lengths):
        f = tmp_path / f'in{k}.pt'
        start = (37 * k) % 100
        rows = block[start:start + n].clone()
        centre = (300 + 7 * k + 3 * torch.arange(n)) % S                       # a peak that wanders: posteriorgram-like
        rows -= ((torch.arange(S)[None, :] - centre[:, None]).abs().float() / 12.0) ** 2
        torch.save(rows.log_softmax(-1).clone(), f)
        ins.append(f)
        outs.append(tmp_path / f'out{k}.pt')
    tf = tmp_path / 'transition.pt'
    x = torch.arange(S)
    tri = torch.clip(87.2 - (x[:, None] - x[None, :]).abs().float(), 0)
    torch.save(tri / tri.sum(dim=1, keepdims=True), tf)
    torbi_amd.from_files_to_files(ins, outs, transition_file=tf, log_probs=True, gpu=0, lengths=lengths)
    trans = torch.load(tf)
    # (which kernel a launch group takes depends on how many batches are resident when the device falls idle: whole tiles from
    # 128 tiles up, clusters below; the same matrix as ONE group of five batches is the band kernel's)
    dev = torch.device('cuda:0')
    matrix = torch.log(trans + torch.finfo(torch.float32).tiny).to(dev)
    rows = [torch.load(ins[k]).to(dev) for k in range(0, 2048)]
    T = max(r.shape[0] for r in rows)
    batches = [torch.zeros((512, T, S), device=dev) for _ in range(4)]
    frames = [torch.tensor(lengths[512 * g:512 * g + 512], dtype=torch.int32, device=dev) for g in range(4)]
    for k, r in enumerate(rows):
        batches[k // 512][k % 512, :r.shape[0]] = torch.log(torch.exp(r) + torch.finfo(torch.float32).tiny)
    prof = []
    got = viterbi.decode_batches(batches, frames, matrix, torch.full((S,), float(np.log(np.float32(1.0 / S) + np.finfo(np.float32).tiny)),
                                                                     device=dev), _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == 'band' and 'true>' in viterbi.last_forward_kernel()       # (either form, the constant's instance)
    for k in range(0, 2048, 97):
        assert torch.equal(got[k // 512][k % 512, :lengths[k]].cpu(), torch.load(outs[k])), f'file {k}'
    for k in range(count):
        a = torch.load(outs[k])
        assert a.dtype == torch.int32 and a.shape == (lengths[k],)
    for k in np.random.default_rng(5).choice(count, size=30, replace=False):
        want = _oracle_for_file(torch.load(ins[k]), trans, S)
        assert np.array_equal(torch.load(outs[k]).numpy(), want), f'file {k} ({lengths[k]} frames)'


@all_paths
def test_grouped_decode_pipeline_equals_the_oracle():
    """DecodePipeline(group=3): batches are collected and decoded three per launch group; a different model, a
    different state count, wait() on a batch that is still being collected, and synchronize() all flush what has
    been collected; every batch equals the oracle."""
    dev = torch.device('cuda:0')
    pipe = torbi_amd.DecodePipeline(dev, depth=2, group=3)
    models = {}
    for S, seed in ((360, 3), (360, 4), (132, 5)):
        models[(S, seed)] = (torch.tensor(synth.scores(2, (S, S), seed=seed), device=dev),
                             torch.tensor(synth.scores(3, (S,), seed=seed), device=dev))
    plan = [((360, 3), 64, 30), ((360, 3), 40, 17), ((360, 3), 96, 45), ((360, 3), 33, 8), ((360, 4), 64, 30),
            ((360, 4), 20, 12), ((132, 5), 50, 9), ((132, 5), 48, 21), ((360, 3), 17, 5)]
    jobs = []
    for k, (key, B, T) in enumerate(plan):
        trans, init = models[key]
        obs = torch.tensor(synth.scores(1, (B, T, key[0]), seed=k), device=dev)
        frames = torch.tensor(np.clip(synth.lengths(B, 1, T, seed=k), 1, T), device=dev)
        jobs.append((obs, frames, trans, init, pipe.decode(obs, frames, trans, init)))
        if k == 3:
            pipe.wait(jobs[-1][4])          # still being collected: flushed and awaited
    pipe.synchronize()
    for obs, frames, trans, init, got in jobs:
        want = oracle.decode(obs.cpu().numpy(), frames.cpu().numpy(), trans.cpu().numpy(), init.cpu().numpy(),
                             num_threads=oracle.max_threads())
        assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('B', [8200, 1000])
def test_time_resident_tile_orders_do_not_change_results(B):
    """The time-resident kernel forms its 16-item tiles from items ranked by length and ranks the tiles across the
    launch group (longest or shortest first); batches above 8192 items keep their order.  None of it may show in
    the indices: ragged lengths, partial last tile, both orders, against the oracle."""
    dev = torch.device('cuda:0')
    T, S = 7, 64
    obs, trans, init = synth.problem(B, T, S, seed=B)
    frames = synth.lengths(B, 1, T, seed=3)
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    d = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    small = _device_problem(37, 5, S, seed=9, dev=dev)
    want_small = oracle.decode(small[0], small[1], trans, init)
    for shortest_first in (False, True):
        got = viterbi.decode_batches([d[0], torch.as_tensor(small[0]).to(dev)], [d[1], torch.as_tensor(small[1]).to(dev)],
                                     d[2], d[3], path='resident', shortest_first=shortest_first)
        np.testing.assert_array_equal(got[0].cpu().numpy(), want)
        np.testing.assert_array_equal(got[1].cpu().numpy(), want_small)


def test_command_line_decodes_files_like_the_reference_cli(tmp_path):
    """`python -m torbi_amd --input_files ... --output_files ... --transition_file ... --log_probs --gpu 0`
    (the reference's flags, torbi/__main__.py:16-49) writes the oracle's decode of every file."""
    import subprocess
    import sys
    S = 48
    gen = torch.Generator().manual_seed(5)
    ins, outs = [], []
    for k, n in enumerate([11, 3, 29]):
        f = tmp_path / f'in{k}.pt'
        torch.save(torch.rand(n, S, generator=gen).log_softmax(-1), f)
        ins.append(str(f))
        outs.append(str(tmp_path / f'out{k}.pt'))
    tf = tmp_path / 'transition.pt'
    torch.save(torch.rand(S, S, generator=gen).softmax(-1), tf)
    root = __import__('conftest').ROOT
    run = subprocess.run([sys.executable, '-m', 'torbi_amd', '--input_files', *ins, '--output_files', *outs,
                          '--transition_file', str(tf), '--log_probs', '--gpu', '0', '--num_threads', '2'],
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    for fin, fout in zip(ins, outs):
        got = torch.load(fout)
        assert got.dtype == torch.int32
        assert np.array_equal(got.numpy(), _oracle_for_file(torch.load(fin), torch.load(tf), S))


def test_auto_leaves_the_time_resident_kernel_when_nothing_is_pruned():
    """Transitions that fall with the prev-state exactly as fast as the posteriors rise make every candidate of a row
    nearly equal: the pruning bound never bites and the time-resident kernel walks every list to its end.  The scan
    statistics of the first AUTO launch group say so (torbi_hip_scan_stats), and later groups with the same matrix
    go to the dense kernel; with the benchmark's inputs the groups stay time-resident.  Same indices either way."""
    dev = torch.device('cuda:0')
    S, T, B, n = 360, 10, 272, 8
    noise_obs, noise_trans, init = synth.problem(B * n, T, S, seed=2)
    ramp = (np.arange(S, dtype=np.float32) * np.float32(0.25))
    cases = {'anti': ((noise_obs * np.float32(2 ** -6) + ramp[None, None, :]).astype(np.float32),
                      (noise_trans * np.float32(2 ** -6) - ramp[None, :]).astype(np.float32), 1),
             'benchmark': (noise_obs, noise_trans, 3)}
    d_init = torch.as_tensor(init).to(dev)
    for name, (obs_all, matrix, later) in cases.items():
        obs_list = [torch.as_tensor(np.ascontiguousarray(obs_all[k * B:(k + 1) * B])).to(dev) for k in range(n)]
        frame_list = [torch.as_tensor(np.clip(synth.lengths(B, 1, T, seed=k), 1, T)).to(dev) for k in range(n)]
        want = {k: oracle.decode(obs_all[k * B:(k + 1) * B], frame_list[k].cpu().numpy(), matrix, init,
                                 num_threads=oracle.max_threads()) for k in (0, n - 1)}
        d_matrix = torch.as_tensor(matrix).to(dev)
        routes = []
        for _ in range(3):
            prof = []
            got = viterbi.decode_batches(obs_list, frame_list, d_matrix, d_init, _profile=prof)
            torch.cuda.synchronize()
            routes.append(int(prof[3]))
            for k in want:
                np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=name)
        assert routes[0] == 3 and routes[-1] == later, (name, routes)


@pytest.mark.parametrize('case', [(1, 40, 1440, None), (9, 30, 360, None), (300, 25, 64, None), (300, 20, 200, None),
                                  (70, 24, 1440, None), (520, 12, 1440, None), (140, 15, 2052, None), (300, 16, 1440, (87, 87)),
                                  (2080, 8, 1440, (87, 87))])
def test_decode_captures_into_a_hip_graph_and_replays_on_new_observations(case):
    """A decode with a caller-owned workspace is a fixed sequence of launches on the caller's stream: torch.cuda.graph captures
    it on every route (one sequence, a handful, small state counts, clusters, whole tiles, both band forms -- the in-launch
    exchanges start from memset nodes), and a replay on NEW observations in the captured buffers decodes those: equal to the
    oracle.  (Eager calls first: the one look at the matrix that synchronises is not taken while capturing, nor are the
    routing statistics read or written.)"""
    B, T, S, band = case
    obs, trans, init = synth.problem(B, T, S, seed=B + S)
    if band is not None:
        trans = _banded(S, *band, seed=3)
    obs2 = synth.problem(B, T, S, seed=B + S + 1)[0]
    frames = synth.lengths(B, 1, T)
    frames[0] = T
    dev = torch.device('cuda:0')
    tobs, tframes, ttrans, tinit = (torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs, frames.astype(np.int32), trans, init))
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    for _ in range(2):
        eager = torbi_amd.decode(tobs, tframes, ttrans, tinit, workspace=ws).clone()
    def reference(o):               # the oracle; the largest shapes against the per-timestep dense route (itself oracle-tested)
        if B * T * S * S < 6e9:
            return oracle.decode(o, frames, trans, init)
        return torbi_amd.decode(torch.as_tensor(o).to(dev), tframes, ttrans, tinit, path='dense').cpu().numpy()
    want = reference(obs)
    np.testing.assert_array_equal(eager.cpu().numpy(), want)
    side = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = torbi_amd.decode(tobs, tframes, ttrans, tinit, workspace=ws)
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    tobs.copy_(torch.as_tensor(obs2))
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), reference(obs2))


# ---- the band kernel (csrc/band_forward.hpp): banded transition matrices, the time loop inside one launch ----------------

def _banded(S, left, right, seed=0):
    """A transition matrix that is finite exactly where -left <= prev - next <= right."""
    _, trans, _ = synth.problem(1, 1, S, seed=seed)
    idx = np.arange(S)
    d = idx[None, :] - idx[:, None]
    return np.where((d >= -left) & (d <= right), trans, -np.inf).astype(np.float32)


def _decode_band(obs, frames, trans, init, want_route='band'):
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs, frames, trans, init)]
    prof = []
    got = torbi_amd.decode(*args, path='band', _profile=prof).cpu().numpy()
    if want_route is not None:
        assert viterbi.ROUTES[int(prof[3])] == want_route, (viterbi.ROUTES[int(prof[3])], want_route)
    return got


@pytest.mark.parametrize('case', [(40, 12, 1440, 87, 87), (17, 9, 360, 10, 3), (64, 25, 360, 22, 22), (100, 7, 1440, 0, 0),
                                  (33, 11, 1024, 5, 60), (16, 6, 1440, 87, 87), (5, 8, 1440, 40, 40), (600, 5, 1440, 87, 87),
                                  (48, 9, 3072, 30, 30), (70, 6, 132, 8, 8), (260, 4, 1444, 86, 88), (24, 10, 512, 100, 100),
                                  (1, 30, 1440, 87, 87), (530, 3, 64, 3, 3), (31, 7, 2048, 120, 0), (96, 5, 768, 0, 150)])
@pytest.mark.parametrize('ties', [False, True])
@pytest.mark.parametrize('form', ['split', 'tile'])
def test_band_kernel_matches_the_oracle(case, ties, form, monkeypatch):
    """torbi_hip_viterbi_decode_banded on matrices that are -inf outside a band (viterbi.cpp:81-104 with -inf candidates never
    winning the strict '>'): one tile and many, ragged lengths, 1 .. 16 members per tile, asymmetric and one-sided bands, a
    band of the diagonal alone, a next-state nothing leads to, coarse grids (many exactly equal candidates: lowest index
    wins), states the members' shares do not divide evenly.  Both forms of the kernel: tiles split over members
    (csrc/band_forward.hpp) and whole tiles with the band streamed from the L2 (csrc/band_tile_forward.hpp; what launch
    groups of at least half a tile per compute unit run, here forced on small groups)."""
    B, T, S, left, right = case
    whole = form == 'tile' and S % 4 == 0 and 64 <= S <= 1536
    monkeypatch.setenv('TORBI_HIP_BAND_FORM', form)
    obs, _, init = synth.problem(B, T, S, seed=B + S)
    trans = _banded(S, left, right, seed=S)
    if ties:
        obs, trans, init = np.round(obs * 2) / 2, np.round(trans * 2) / 2, np.round(init)
    trans[S // 3] = -np.inf
    frames = np.clip(synth.lengths(B, 1, T, seed=3), 1, T).astype(np.int32)
    frames[0] = T
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    np.testing.assert_array_equal(_decode_band(obs, frames, trans, init), want)
    assert ('band_tile_kernel' in viterbi.last_forward_kernel()) == whole, viterbi.last_forward_kernel()


@pytest.mark.parametrize('width', [1, 7, 44, 87, 88, 89, 93, 122, 123, 254, 300])
@pytest.mark.parametrize('form', ['split', 'tile'])
def test_band_width_sweep_at_1440_states(width, form, monkeypatch):
    """Half widths from the diagonal alone over the reference's pitch model (reach 87 either way, torbi/evaluate/core.py:24-33)
    to the widest band 1440 states allow (split form: reach 121, eleven members of 132 next-states; whole tiles: any reach up
    to 254, the backtrace's window), and bands the kernel does not cover (the call then is
    torbi_hip_viterbi_decode_batches); a dead row; ties."""
    monkeypatch.setenv('TORBI_HIP_BAND_FORM', form)
    B, T, S = 40, 7, 1440
    obs, trans, init = synth.problem(B, T, S, seed=width)
    obs = np.round(obs * 2) / 2
    trans = np.round(trans * 2) / 2
    idx = np.arange(S)
    trans = np.where(np.abs(idx[:, None] - idx[None, :]) < width, trans, -np.inf).astype(np.float32)
    trans[S // 3] = -np.inf
    frames = np.clip(synth.lengths(B, 1, T, seed=2), 1, T).astype(np.int32)
    frames[0] = T
    want = oracle.decode(obs.astype(np.float32), frames, trans, init)
    reach = width - 1
    covered = torbi_amd._lib.load().torbi_hip_band_members(B, S, reach, reach, 0) > 0
    assert covered == (reach <= (254 if form == 'tile' else 121))
    got = _decode_band(obs.astype(np.float32), frames, trans, init, want_route='band' if covered else None)
    np.testing.assert_array_equal(got, want)


def test_auto_takes_the_band_kernel_for_the_pitch_transition():
    """The reference's own evaluation workload (torbi/evaluate/core.py:24-33): peaked posteriorgram-like rows, the triangular
    pitch band, ragged 512 x 500 x 1440 -- AUTO routes it to the band kernel (one batch and a launch group), the first 64
    items equal the oracle's, and every named path agrees on all of them."""
    import math
    dev = torch.device('cuda:0')
    B, T, S = 512, 500, 1440
    gen = torch.Generator(device=dev).manual_seed(11)
    logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
    centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
    logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
    peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
    del logits
    band_np = synth.banded_transition(S, 87.2)
    band = torch.from_numpy(band_np).to(dev)
    init_np = np.full((S,), math.log(1.0 / S), np.float32)
    init = torch.from_numpy(init_np).to(dev)
    frames_np = np.clip(synth.lengths(B, 300, T, seed=5), 1, T).astype(np.int32)
    frames_np[0] = T
    frames = torch.from_numpy(frames_np).to(dev)
    assert viterbi.band_reach(band, band, S) == (87, 87)
    prof = []
    got = torbi_amd.decode(peaked, frames, band, init, _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == 'band'
    got_np = got.cpu().numpy()
    want = oracle.decode(peaked[:64].cpu().numpy(), frames_np[:64], band_np, init_np, num_threads=oracle.max_threads())
    np.testing.assert_array_equal(got_np[:64], want)
    for path in ('dense', 'cluster'):
        np.testing.assert_array_equal(torbi_amd.decode(peaked, frames, band, init, path=path).cpu().numpy(), got_np)
    group = viterbi.decode_batches([peaked, peaked[:100], peaked[:17]], [frames, frames[:100], frames[:17]], band, init,
                                   _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == 'band'
    for g, n in zip(group, (B, 100, 17)):
        np.testing.assert_array_equal(g.cpu().numpy(), got_np[:n])
    # the same ragged batch eight times in one call: 256 tiles = WHOLE tiles (csrc/band_tile_forward.hpp), one forward launch
    eight = viterbi.decode_batches([peaked] * 8, [frames] * 8, band, init, _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == 'band' and int(prof[2]) == 1 and 'band_tile_kernel<2, 12, false>' in viterbi.last_forward_kernel()
    for g in eight:
        np.testing.assert_array_equal(g.cpu().numpy(), got_np)
    # ... and with the matrix the reference's evaluation really decodes with, log(p + tiny): against the dense kernel
    tiny_band = torch.from_numpy(synth.banded_transition(S, 87.2, tiny=True)).to(dev)
    eight = viterbi.decode_batches([peaked] * 8, [frames] * 8, tiny_band, init, _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == 'band' and 'band_tile_kernel<2, 12, true>' in viterbi.last_forward_kernel()
    dense = torbi_amd.decode(peaked, frames, tiny_band, init, path='dense').cpu().numpy()
    for g in eight:
        np.testing.assert_array_equal(g.cpu().numpy(), dense)
    stats = viterbi.scan_stats(torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev), B, T, S)
    del stats


def _peaked_pitch_problem(B, T, S, seed):
    """Posteriorgram-like rows and the reference's triangular pitch band (torbi/evaluate/core.py:24-33), on the device."""
    import math
    dev = torch.device('cuda:0')
    gen = torch.Generator(device=dev).manual_seed(seed)
    logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
    centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
    logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
    peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
    band_np = synth.banded_transition(S, 87.2)
    init_np = np.full((S,), math.log(1.0 / S), np.float32)
    return peaked, band_np, init_np


@pytest.mark.parametrize('waves', [None, '8'])
def test_band_launch_group_runs_whole_tiles(waves, monkeypatch):
    """A launch group with a tile for at least every other compute unit runs the band kernel's WHOLE-TILE form
    (csrc/band_tile_forward.hpp: one workgroup per 16-item tile, the band streamed from the L2, no hand-off between
    workgroups) under AUTO: eight ragged batches of the reference's pitch workload (torbi/evaluate/core.py:24-33), sizes that
    leave partial tiles.  Every batch equals the dense kernel's decode of it (another kernel, every cell), 48 items spread
    over the batches equal the oracle's, nothing gave up, and the group took ONE forward launch.  Also with eight waves
    per workgroup (three blocks per wave: the instance wide bands run)."""
    if waves:
        monkeypatch.setenv('TORBI_HIP_TILE_WAVES', waves)
    dev = torch.device('cuda:0')
    T, S = 40, 1440
    sizes = [512, 512, 300, 512, 100, 17, 512, 129]
    peaked, band_np, init_np = _peaked_pitch_problem(512, T, S, seed=23)
    band, init = torch.from_numpy(band_np).to(dev), torch.from_numpy(init_np).to(dev)
    obs_list, frame_list, frames_np = [], [], []
    for k, B in enumerate(sizes):
        f = np.clip(synth.lengths(B, 1, T, seed=40 + k), 1, T).astype(np.int32)
        f[0] = T
        frames_np.append(f)
        obs_list.append(torch.roll(peaked, k, dims=0)[:B].contiguous())
        frame_list.append(torch.from_numpy(f).to(dev))
    assert sum((B + 15) // 16 for B in sizes) >= 128
    prof = []
    got = viterbi.decode_batches(obs_list, frame_list, band, init, _profile=prof)
    torch.cuda.synchronize()
    assert viterbi.ROUTES[int(prof[3])] == 'band' and int(prof[2]) == 1
    assert 'band_tile_kernel<2, 12, false>' in viterbi.last_forward_kernel() or (waves and 'band_tile_kernel<3, 8, false>' in viterbi.last_forward_kernel())
    for k, B in enumerate(sizes):
        dense = torbi_amd.decode(obs_list[k], frame_list[k], band, init, path='dense')
        np.testing.assert_array_equal(got[k].cpu().numpy(), dense.cpu().numpy(), err_msg=f'batch {k}')
        pick = np.random.default_rng(k).choice(B, size=6, replace=False)
        want = oracle.decode(obs_list[k][pick].cpu().numpy(), frames_np[k][pick], band_np, init_np, num_threads=oracle.max_threads())
        np.testing.assert_array_equal(got[k].cpu().numpy()[pick], want, err_msg=f'batch {k}')


def _decode_banded_over(obs, frames, trans, init, left, right, background):
    """torbi_hip_viterbi_decode_banded_over through ctypes with the band and the constant as the caller's promise (BAND named)."""
    import ctypes
    from torbi_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    B, T, S = obs.shape
    o, f, m, i = (torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs, frames, trans, init))
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    out = torch.empty((B, T), dtype=torch.int32, device=dev)
    one = (_lib.Batch * 1)(_lib.Batch(o.data_ptr(), f.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, T))
    phases = (ctypes.c_float * 6)()
    rc = lib.torbi_hip_viterbi_decode_banded_over(one, 1, m.data_ptr(), i.data_ptr(), S, left, right, ctypes.c_float(background), 0,
                                                  ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                                  viterbi._path_flag('band'), phases)
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu().numpy(), viterbi.ROUTES[int(phases[3])], ws


@pytest.mark.parametrize('case', [(300, 9, 360, 10, 3, -3.0), (300, 7, 360, 22, 22, -40.0), (272, 8, 1440, 87, 87, -87.33654),
                                  (260, 6, 1024, 5, 60, -1.0), (300, 9, 360, 10, 10, 2.5)])
@pytest.mark.parametrize('rows', ['random', 'peaked', 'ties'])
@pytest.mark.parametrize('form', ['tile', 'split'])
def test_band_with_a_constant_outside_matches_the_oracle(case, rows, form, monkeypatch):
    """The reference's evaluation decodes with log(p + tiny) (torbi/evaluate/core.py:97-103 -> torbi/core.py:341-347): ONE
    constant outside the band, not -inf.  The whole-tile band kernel decides every output exactly from the band and the row's
    maximum (csrc/band_tile_forward.hpp): background values from "never matters" (-87.3 under random rows) to "wins almost
    everywhere" (-1, +2.5: above the in-band entries), peaked rows whose tails sit far below the peak (candidates from outside
    the band win wherever the peak is out of reach), coarse grids (ties between a candidate inside and one outside the band:
    lowest index wins), ragged lengths, a next-state nothing inside the band leads to.  AUTO finds band and constant itself.
    Both forms: whole tiles, and tiles split over members that exchange their rows' maxima (csrc/band_forward.hpp, <true>)."""
    monkeypatch.setenv('TORBI_HIP_BAND_FORM', form)
    B, T, S, left, right, c = case
    obs, trans, init = synth.problem(B, T, S, seed=B + S)
    idx = np.arange(S)
    d = idx[None, :] - idx[:, None]
    inside = (d >= -left) & (d <= right)
    if rows == 'peaked':
        centre = synth.lengths(B * T, 0, S - 1, seed=5).reshape(B, T, 1)
        obs = (-np.abs(idx[None, None, :] - centre) * 1.5 + obs * 0.25).clip(min=-87.0).astype(np.float32)
    if rows == 'ties':
        obs, trans, init = np.round(obs * 2) / 2, np.round(trans * 2) / 2, np.round(init)
    trans = np.where(inside, trans, np.float32(c)).astype(np.float32)
    if rows == 'ties':          # a next-state nothing inside the band leads to (the constant stays outside; AUTO then leaves the matrix alone)
        trans[S // 3, max(0, S // 3 - left):S // 3 + right + 1] = -np.inf
    frames = np.clip(synth.lengths(B, 1, T, seed=3), 1, T).astype(np.int32)
    frames[0] = T
    want = oracle.decode(obs.astype(np.float32), frames, trans, init, num_threads=oracle.max_threads())
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs.astype(np.float32), frames, trans, init)]
    # the entry point with the band and the constant as the caller's promise -- whatever the constant
    got, route, _ = _decode_banded_over(obs.astype(np.float32), frames, trans, init, left, right, c)
    assert route == 'band'
    assert ('band_tile_kernel' if form == 'tile' else 'band_forward_kernel<true>') in viterbi.last_forward_kernel()
    np.testing.assert_array_equal(got, want)
    # AUTO takes a finite constant only from a matrix whose every other entry lies above it (a pitch matrix does: log(tiny)
    # against -10.5 and more inside the band): where the band reaches down to the constant, the outputs next to a row's maximum
    # cannot be decided from the maximum alone
    above = bool((trans[inside] > c).all())
    assert (viterbi.band_over(args[2], args[2], S) == (left, right, pytest.approx(c))) == above
    prof = []
    got = torbi_amd.decode(*args, _profile=prof)
    assert (viterbi.ROUTES[int(prof[3])] == 'band') == above
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_the_reference_evaluation_matrix_runs_on_the_band_kernel():
    """torbi.evaluate calls from_files_to_files(transition_file=<pitch matrix>, log_probs=True) (torbi/evaluate/core.py:97-103):
    the matrix the operator sees is log(p + tiny) -- log(tiny) outside the band (torbi/core.py:341-347).  A launch group of five
    ragged batches of posteriorgram-like rows under AUTO: the whole-tile band kernel (two batches: the split form), every batch equal to the dense kernel's
    decode (every cell of the matrix), 24 items equal to the oracle's; rows with network-like tails (nothing outside the band
    ever wins) and rows clamped at log(tiny) (candidates from outside the band win wherever the last peak is out of reach)."""
    import math
    dev = torch.device('cuda:0')
    T, S = 40, 1440
    tiny = torch.finfo(torch.float32).tiny
    x = np.arange(S)
    tri = np.clip(87.2 - np.abs(x[:, None] - x[None, :]), 0, None).astype(np.float32)
    probs = torch.from_numpy(tri / tri.sum(axis=1, keepdims=True)).to(dev)
    band = torch.log(probs + tiny)
    band_np = band.cpu().numpy()
    left, right, c = viterbi.band_over(band, band, S)
    assert (left, right) == (87, 87) and c == pytest.approx(math.log(tiny)) and viterbi.band_reach(band, band, S) is None
    init_np = np.full((S,), math.log(1.0 / S), np.float32)
    init = torch.from_numpy(init_np).to(dev)
    peaked, _, _ = _peaked_pitch_problem(512, T, S, seed=31)                 # tails clamped at log(tiny)
    soft = torch.log_softmax(peaked.clamp(min=-30.0), dim=-1)               # tails like a network's softmax
    for name, rows in (('clamped', peaked), ('soft', soft)):
        # (two batches: tiles split over members that exchange their rows' maxima, band_forward_kernel<true>)
        few = viterbi.decode_batches([rows[:300].contiguous(), rows[300:400].contiguous()],
                                     [torch.full((300,), T, dtype=torch.int32, device=dev), torch.full((100,), T - 3, dtype=torch.int32, device=dev)],
                                     band, init)
        assert 'band_forward_kernel<true>' in viterbi.last_forward_kernel(), name
        for g, (lo, hi, f) in zip(few, ((0, 300, T), (300, 400, T - 3))):
            dense = torbi_amd.decode(rows[lo:hi].contiguous(), torch.full((hi - lo,), f, dtype=torch.int32, device=dev), band, init, path='dense')
            np.testing.assert_array_equal(g.cpu().numpy(), dense.cpu().numpy(), err_msg=f'{name} {lo}')
        sizes = [512, 512, 512, 512, 500]
        obs_list, frame_list, frames_np = [], [], []
        for k, B in enumerate(sizes):
            f = np.clip(synth.lengths(B, 1, T, seed=60 + k), 1, T).astype(np.int32)
            f[0] = T
            frames_np.append(f)
            obs_list.append(torch.roll(rows, 3 * k, dims=0)[:B].contiguous())
            frame_list.append(torch.from_numpy(f).to(dev))
        prof = []
        got = viterbi.decode_batches(obs_list, frame_list, band, init, _profile=prof)
        torch.cuda.synchronize()
        assert viterbi.ROUTES[int(prof[3])] == 'band' and 'band_tile_kernel' in viterbi.last_forward_kernel(), name
        for k, B in enumerate(sizes):
            dense = torbi_amd.decode(obs_list[k], frame_list[k], band, init, path='dense')
            np.testing.assert_array_equal(got[k].cpu().numpy(), dense.cpu().numpy(), err_msg=f'{name} batch {k}')
            pick = np.random.default_rng(k).choice(B, size=6, replace=False)
            want = oracle.decode(obs_list[k][pick].cpu().numpy(), frames_np[k][pick], band_np, init_np, num_threads=oracle.max_threads())
            np.testing.assert_array_equal(got[k].cpu().numpy()[pick], want, err_msg=f'{name} batch {k}')
        space = torch.empty(viterbi.workspace_bytes(512, T, S), dtype=torch.uint8, device=dev)
        del space


def test_band_constant_that_the_tests_cannot_decide_is_decoded_in_the_reference_order():
    """In-band entries BELOW the constant outside the band, exactly where a row's maximum stands: the candidates from outside can
    win although the maximum itself is in reach, and the kernel's two tests decide nothing -- the batch raises its alarm and is
    decoded again in the reference's order (csrc/nonfinite.hpp).  Indices equal the oracle's."""
    B, T, S, reach, c = 264, 6, 360, 12, -2.0
    obs, trans, init = synth.problem(B, T, S, seed=91)
    idx = np.arange(S)
    inside = np.abs(idx[None, :] - idx[:, None]) <= reach
    trans = np.where(inside, trans - 12.0, np.float32(c)).astype(np.float32)       # the band far BELOW the constant
    frames = np.full((B,), T, np.int32)
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    import os
    for form in ('tile', 'split'):
        os.environ['TORBI_HIP_BAND_FORM'] = form
        try:
            got, route, ws = _decode_banded_over(obs, frames, trans, init, reach, reach, c)
        finally:
            del os.environ['TORBI_HIP_BAND_FORM']
        assert route == 'band', form
        np.testing.assert_array_equal(got, want, err_msg=form)


def test_band_launch_never_holds_more_members_than_are_resident():
    """Round-5 advisor: the members of a tile wait for each other INSIDE a launch, so a launch must not hold more members
    than the chip has units for (a dispatch class of R x ceil(tiles / 8) workgroups runs on one XCD: cus / 8 units).  600
    sequences with reach 88 need nine members per tile: 38 tiles x 9 = 342 workgroups used to be launched at once -- the
    late ones spun for their wait budget and the repair launch decoded their tiles again.  Now the group is decoded 24 tiles
    per launch (two launches), nothing gives up, and the result is the dense kernel's."""
    dev = torch.device('cuda:0')
    B, T, S, reach = 600, 500, 1440, 88
    peaked, _, init_np = _peaked_pitch_problem(B, T, S, seed=5)
    idx = np.arange(S)
    trans_np = np.where(np.abs(idx[:, None] - idx[None, :]) <= reach, synth.scores(synth.STREAM_TRANSITION, (S, S), seed=3), -np.inf)
    trans = torch.from_numpy(trans_np.astype(np.float32)).to(dev)
    init = torch.from_numpy(init_np).to(dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    assert torbi_amd._lib.load().torbi_hip_band_members(B, S, reach, reach, 0) == 9
    prof = []
    got = torbi_amd.decode(peaked, frames, trans, init, workspace=space, path='band', _profile=prof)
    torch.cuda.synchronize()
    assert viterbi.ROUTES[int(prof[3])] == 'band' and int(prof[2]) == 2
    assert int(viterbi.scan_stats(space, B, T, S).cpu()[127]) == 0
    assert prof[0] < 60.0, f'forward took {prof[0]:.1f} ms: members waiting for units?'
    np.testing.assert_array_equal(got.cpu().numpy(), torbi_amd.decode(peaked, frames, trans, init, path='dense').cpu().numpy())


@pytest.mark.parametrize('shape', [(40, 12, 360, 10), (17, 9, 1440, 87), (130, 7, 724, 30), (40, 12, 360, 10, -40.0), (130, 7, 724, 30, -87.33654)])
def test_band_launch_that_gives_up_waiting_is_repaired(shape, monkeypatch):
    """The members of a tile wait for each other's halo rows inside the launch; every wait is bounded.  With a budget of 0
    every failed poll gives up: the members flag their tile and band_repair_kernel decodes it again without hand-offs --
    indices and final posterior rows are the oracle's, the give-ups are counted; without the limit nothing gives up."""
    B, T, S, reach = shape[:4]
    obs, _, init = synth.problem(B, T, S, seed=41)
    trans = _banded(S, reach, reach, seed=7)
    if len(shape) > 4:          # ONE constant outside the band (band_forward_kernel<true>; band_repair_kernel evaluates it too)
        trans = np.where(np.isneginf(trans), np.float32(shape[4]), trans).astype(np.float32)
        monkeypatch.setenv('TORBI_HIP_BAND_FORM', 'split')
    frames = np.clip(synth.lengths(B, 1, T, seed=6), 1, T).astype(np.int32)
    frames[0] = T
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)]
    space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    want, want_post = oracle.decode(obs, frames, trans, init, return_posterior=True)
    for limit, gave_up in (('0', True), (None, False), ('0', True)):
        if limit is None:
            monkeypatch.delenv('TORBI_HIP_CLUSTER_WAIT_US', raising=False)
        else:
            monkeypatch.setenv('TORBI_HIP_CLUSTER_WAIT_US', limit)
        prof = []
        got = torbi_amd.decode(*args, workspace=space, path='band', _profile=prof)
        assert viterbi.ROUTES[int(prof[3])] == 'band'
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        post = viterbi.read_posterior(space, args[1], B, T, S).cpu().numpy()
        assert np.array_equal(post.view(np.uint32), want_post.view(np.uint32))
        stats = viterbi.scan_stats(space, B, T, S).cpu()
        assert (int(stats[127]) > 0) == gave_up, (limit, int(stats[127]))


def test_band_entry_point_falls_back_and_validates():
    """torbi_hip_viterbi_decode_banded through ctypes: a band it does not cover, a path other than AUTO / BAND and a
    misaligned matrix all decode as torbi_hip_viterbi_decode_batches would (same indices); negative reaches are EINVAL."""
    import ctypes
    from torbi_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    B, T, S = 48, 6, 360
    obs, _, init = synth.problem(B, T, S, seed=2)
    trans = _banded(S, 9, 9, seed=3)
    frames = np.full((B,), T, np.int32)
    want = oracle.decode(obs, frames, trans, init)
    o, f, i = (torch.as_tensor(x).to(dev) for x in (obs, frames, init))
    padded = torch.zeros((S * S + 1,), dtype=torch.float32, device=dev)
    padded[1:] = torch.as_tensor(trans).reshape(-1).to(dev)
    aligned = torch.as_tensor(trans).to(dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    out = torch.empty((B, T), dtype=torch.int32, device=dev)
    one = (_lib.Batch * 1)(_lib.Batch(o.data_ptr(), f.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, T))
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    phases = (ctypes.c_float * 6)()
    for matrix, left, right, flags, route in ((aligned, 9, 9, viterbi._path_flag('band'), 'band'),
                                              (aligned, 9, 9, viterbi._path_flag('auto'), 'band'),
                                              (aligned, 9, 9, viterbi._path_flag('dense'), 'dense'),
                                              (aligned, 300, 300, viterbi._path_flag('band'), 'cluster'),
                                              (padded[1:], 9, 9, viterbi._path_flag('band'), 'cluster')):
        out.zero_()
        rc = lib.torbi_hip_viterbi_decode_banded(one, 1, matrix.data_ptr(), i.data_ptr(), S, left, right, 0, stream, flags, phases)
        assert rc == 0
        assert viterbi.ROUTES[int(phases[3])] == route, (viterbi.ROUTES[int(phases[3])], route)
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    assert lib.torbi_hip_viterbi_decode_banded(one, 1, aligned.data_ptr(), i.data_ptr(), S, -1, 9, 0, stream, 0, None) == -1
    left, right = ctypes.c_int(-1), ctypes.c_int(-1)
    assert lib.torbi_hip_band_reach(aligned.data_ptr(), S, 0, stream, ctypes.byref(left), ctypes.byref(right)) == 0
    assert (left.value, right.value) == (9, 9)
    lop = torch.as_tensor(_banded(S, 4, 31, seed=1)).to(dev)
    assert lib.torbi_hip_band_reach(lop.data_ptr(), S, 0, stream, ctypes.byref(left), ctypes.byref(right)) == 0
    assert (left.value, right.value) == (4, 31)


# ---- NaN and +inf inputs (csrc/nonfinite.hpp): the reference's results on every route ----------------------------------------

def _poison(kind, obs, trans, init, frames):
    """The NaN cases of tests/test_oracle.py (where the oracle is checked against the reference operator itself), placed
    where the shape allows; 'inf': +inf observations that meet -inf transitions as NaN candidates at prev-state 0 (they
    stay, viterbi.cpp:94-100) and elsewhere (they lose)."""
    B, T, S = obs.shape
    nan = np.float32('nan')
    obs, trans, init = obs.copy(), trans.copy(), init.copy()
    b1, b2 = min(1, B - 1), min(2, B - 1)
    if kind == 'observation':
        obs[0, min(3, T - 1), 5 % S] = nan
    elif kind == 'matrix':
        trans[7 % S, 0] = nan
        trans[9 % S, 11 % S] = nan
    elif kind == 'initial':
        init[0] = nan
    elif kind == 'final_row':
        obs[b1, frames[b1] - 1, 17 % S] = nan
        obs[b2, 0, :] = nan
    elif kind == 'inf':
        obs[0, min(2, T - 1), 0] = np.inf
        obs[b1, min(4, T - 1), 3 % S] = np.inf
        trans[6 % S, 0] = -np.inf
        trans[5 % S, 3 % S] = -np.inf
    return obs, trans, init


NONFINITE_ROUTES = [  # (B, T, S, path, environment, route expected)
    (3, 12, 40, 'auto', {}, 'small'),                                   # one wavefront per sequence, backpointers
    (600, 9, 40, 'auto', {'TORBI_HIP_SMALL_VALUE': '1'}, 'small'),       # ... value-only
    (9, 10, 200, 'auto', {}, 'small'),                                   # one workgroup per sequence
    (2, 14, 1440, 'auto', {}, 'held'),
    (4, 9, 1440, 'auto', {}, 'generic'),
    (8, 9, 300, 'auto', {}, 'rows'),
    (40, 10, 360, 'auto', {}, 'cluster'),
    (40, 10, 360, 'resident', {}, 'resident'),
    (64, 8, 360, 'dense', {}, 'dense'),
]


@pytest.mark.parametrize('kind', ['observation', 'matrix', 'initial', 'final_row', 'inf', 'clean'])
@pytest.mark.parametrize('shape', NONFINITE_ROUTES, ids=lambda s: f'{s[0]}x{s[1]}x{s[2]}-{s[3]}')
def test_nan_and_inf_inputs_decode_as_the_reference_operator(shape, kind, monkeypatch):
    """The reference is deterministic on NaN (viterbi.cpp:94-100 never replaces a NaN candidate at prev-state 0 and never lets
    one win elsewhere; :218 ATen's argmax takes the first NaN of the final row) and +inf meets -inf as NaN; the oracle
    restates exactly that and is pinned against the reference operator on these cases (tests/test_oracle.py).  Every route
    of the HIP path must give the same indices: its kernels raise an alarm when they produce a NaN / +inf posterior value
    (or find one in the matrix / the observations), and the items that read one are decoded again exactly as the reference
    does it (csrc/nonfinite.hpp).  'clean': the same shapes without any, through the same launches."""
    B, T, S, path, env, route = shape
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    obs, trans, init = synth.problem(B, T, S, seed=3000 + B + S)
    frames = np.clip(synth.lengths(B, 2, T, seed=9), 2, T).astype(np.int32)
    frames[0] = T
    if kind != 'clean':
        obs, trans, init = _poison(kind, obs, trans, init, frames)
    want = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    dev = torch.device('cuda:0')
    args = [torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs, frames, trans, init)]
    prof = []
    got = torbi_amd.decode(*args, path=path, _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == route
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('kind', ['observation', 'matrix', 'initial', 'final_row', 'inf'])
@pytest.mark.parametrize('form', ['split', 'tile'])
def test_nan_and_inf_inputs_on_the_band_kernels(kind, form, monkeypatch):
    """... and a banded matrix through both forms of the band kernel, as one batch and as a launch group of two batches of which
    only the second reads a NaN."""
    monkeypatch.setenv('TORBI_HIP_BAND_FORM', form)
    B, T, S, reach = 40, 10, 360, 10
    obs, _, init = synth.problem(B, T, S, seed=77)
    trans = _banded(S, reach, reach, seed=5)
    frames = np.clip(synth.lengths(B, 2, T, seed=4), 2, T).astype(np.int32)
    frames[0] = T
    clean = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads())
    bad_obs, bad_trans, bad_init = _poison(kind, obs, trans, init, frames)
    if kind in ('matrix', 'inf'):
        bad_trans = np.where(np.isneginf(trans), -np.inf, bad_trans).astype(np.float32)      # (the band stays a band)
    want = oracle.decode(bad_obs, frames, bad_trans, bad_init, num_threads=oracle.max_threads())
    dev = torch.device('cuda:0')
    to = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
    prof = []
    got = torbi_amd.decode(to(bad_obs), to(frames), to(bad_trans), to(bad_init), path='band', _profile=prof)
    assert viterbi.ROUTES[int(prof[3])] == 'band'
    assert ('band_tile_kernel' in viterbi.last_forward_kernel()) == (form == 'tile')
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    if kind in ('observation', 'final_row'):       # a group: the matrix is shared, only the second batch reads a NaN
        both = viterbi.decode_batches([to(obs), to(bad_obs)], [to(frames), to(frames)], to(trans), to(init), path='band')
        np.testing.assert_array_equal(both[0].cpu().numpy(), clean)
        np.testing.assert_array_equal(both[1].cpu().numpy(), want)


@pytest.mark.parametrize('kind', ['observation', 'initial', 'final_row', 'first'])
@pytest.mark.parametrize('probabilities', [False, True])
def test_nan_inputs_on_the_uniform_transition_entry(kind, probabilities):
    """The reference's default call (transition=None: every entry log(1 / S), torbi/core.py:175-180) through
    torbi_hip_viterbi_decode_uniform[_probabilities]: an item that reads a NaN is decoded again inside the same launch in the
    reference's order of evaluation (csrc/uniform_decode.hpp, faithful_uniform_item) -- against the oracle on the materialised
    matrix."""
    import math
    B, T, S = 5, 11, 300
    rng = np.random.default_rng(12)
    probs = rng.random((B, T, S)).astype(np.float32) + 1e-3
    probs /= probs.sum(axis=-1, keepdims=True)
    init_p = (rng.random(S).astype(np.float32) + 1e-3)
    init_p /= init_p.sum()
    frames = np.array([T, T - 3, T, 2, T], np.int32)
    nan = np.float32('nan')
    if kind == 'observation':
        probs[0, 4, 7] = nan
    elif kind == 'initial':
        init_p[0] = nan
    elif kind == 'final_row':
        probs[1, frames[1] - 1, 17] = nan
        probs[2, 0, :] = nan
    else:
        probs[4, 3, 0] = nan                   # prev-state 0 of a middle row: the NaN stays for the rest of the item
    dev = torch.device('cuda:0')
    c = float(torch.tensor(math.log(1.0 / S), dtype=torch.float32))
    tiny = torch.finfo(torch.float32).tiny
    scores = torch.log(torch.exp(torch.log(torch.from_numpy(probs).to(dev))) + tiny)       # core.py:189-197, on this device
    init = torch.log(torch.from_numpy(init_p)).to(dev)
    want = oracle.decode(scores.cpu().numpy(), frames, np.full((S, S), c, np.float32), init.cpu().numpy(),
                         num_threads=oracle.max_threads())
    f = torch.from_numpy(frames).to(dev)
    if probabilities:
        got = torbi_amd.decode_uniform(torch.from_numpy(probs).to(dev), f, c, init, probabilities=True)
    else:
        got = torbi_amd.decode_uniform(scores, f, c, init)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_a_stale_band_promise_is_caught_on_the_device():
    """Round-5 advisor: torbi_amd.decode keeps the band of a matrix with the tensor's notes, keyed on its version counter; an edit
    that does not bump the counter (`.data`, memory shared with numpy, another library's kernel) leaves a stale promise, and the
    band kernels never read outside the promised band.  The launch that looks at the matrix for NaN (csrc/nonfinite.hpp) also
    checks the promise: a finite entry outside the band raises the alarm and every item is decoded again on the WHOLE matrix.
    Here the entry point is called with reach 9 for a matrix whose band was widened to 40 behind its back."""
    import ctypes
    from torbi_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    B, T, S = 48, 6, 360
    obs, _, init = synth.problem(B, T, S, seed=2)
    frames = np.full((B,), T, np.int32)
    o, f, i = (torch.as_tensor(x).to(dev) for x in (obs, frames, init))
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    out = torch.empty((B, T), dtype=torch.int32, device=dev)
    one = (_lib.Batch * 1)(_lib.Batch(o.data_ptr(), f.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, T))
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    phases = (ctypes.c_float * 6)()
    for reach in (9, 40):
        trans = _banded(S, reach, reach, seed=3)
        matrix = torch.as_tensor(trans).to(dev)
        out.zero_()
        rc = lib.torbi_hip_viterbi_decode_banded(one, 1, matrix.data_ptr(), i.data_ptr(), S, 9, 9, 0, stream, viterbi._path_flag('band'), phases)
        assert rc == 0 and viterbi.ROUTES[int(phases[3])] == 'band'
        np.testing.assert_array_equal(out.cpu().numpy(), oracle.decode(obs, frames, trans, init), err_msg=f'true reach {reach}')


def test_generic_route_takes_more_items_than_one_grid_dimension_holds():
    """The per-timestep trellis kernels index the item by gridDim.y (at most 65535): a batch of 70 000 sequences over ONE
    state (AUTO: generic -- nothing else covers S == 1) and over 8 states with DENSE named (generic below 64 states) used to
    fail with hipErrorInvalidConfiguration once the 64 x 64 tile kernel was gone (round-4 advisor); now a timestep is
    launched in slices.  One state: every index is 0; eight states: the oracle's on a sample, every row within range."""
    dev = torch.device('cuda:0')
    B, T = 70000, 3
    for S, path in ((1, 'auto'), (8, 'dense')):
        obs, trans, init = synth.problem(B, T, S, seed=9)
        frames = np.full((B,), T, np.int32)
        prof = []
        got = torbi_amd.decode(*[torch.as_tensor(x).to(dev) for x in (obs, frames, trans, init)], path=path, _profile=prof)
        assert viterbi.ROUTES[int(prof[3])] == 'generic'
        got = got.cpu().numpy()
        pick = np.r_[0:40, 65500:65600, B - 40:B]
        want = oracle.decode(obs[pick], frames[pick], trans, init)
        np.testing.assert_array_equal(got[pick], want)
        assert got.min() >= 0 and got.max() < S


def test_auto_gates_follow_the_data_not_the_first_call():
    """AUTO's data-dependent gates (torbi_amd/viterbi.py: scan depth of the time-resident kernel against the dense kernel) used
    to be measured once per matrix: peaked batches first, flat ones later left the matrix on clusters through a 3-4x cliff,
    the reverse on the dense kernel for ever (round-4 review).  Now every time-resident launch AUTO chose leaves a sample,
    the depth is a mean that leans on the newest one, and a matrix the gates keep on the dense kernel is looked at again every
    third call.  Peaked batches, then flat ones, then peaked ones with ONE transition tensor: the route changes within three
    calls each time, and every call's indices are the oracle's."""
    dev = torch.device('cuda:0')
    B, T, S = 256, 10, 1440
    _, trans, init = synth.problem(1, 1, S, seed=3)
    rng = np.random.default_rng(8)
    # Eight prev-states nothing likes to come from (their columns hold the matrix's minimum).  Rows that put their largest
    # posteriors THERE defeat the bound: the seeds' candidates are poor, every other posterior is far below the threshold,
    # and a scan runs down its list until the entries themselves are as poor (~85 of 90 blocks).  Flat rows are pruned after
    # ~6 blocks by the matrix's own spread.
    bad = np.arange(8) * 170 + 40
    trans = trans.copy()
    trans[:, bad] = np.float32(-16.0)
    peaked = (rng.integers(0, 2, size=(B, T, S)) * np.float32(2.0 ** -12)).astype(np.float32)      # "peaked" = prunable
    flat = peaked.copy()
    flat[:, :, bad] += np.float32(15.0)                                                           # "flat" = nothing to prune
    frames = np.full((B,), T, np.int32)
    d_trans, d_init, d_frames = (torch.as_tensor(x).to(dev) for x in (trans, init, frames))
    want = {'flat': oracle.decode(flat, frames, trans, init, num_threads=oracle.max_threads()),
            'peaked': oracle.decode(peaked, frames, trans, init, num_threads=oracle.max_threads())}
    data = {'flat': torch.as_tensor(flat).to(dev), 'peaked': torch.as_tensor(peaked).to(dev)}
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    torbi_amd.reset_path_state()

    def calls_until(kind, route, limit):
        for n in range(1, limit + 1):
            prof = []
            got = torbi_amd.decode(data[kind], d_frames, d_trans, d_init, workspace=ws, _profile=prof)
            np.testing.assert_array_equal(got.cpu().numpy(), want[kind])
            if viterbi.ROUTES[int(prof[3])] == route:
                return n
        return None

    assert calls_until('peaked', 'cluster', 1) == 1                     # nothing known: clusters
    for _ in range(2):
        assert calls_until('peaked', 'cluster', 1) == 1                 # shallow scans: stays
    assert calls_until('flat', 'dense', 3) is not None                  # flat rows: the dense kernel within three calls
    assert calls_until('flat', 'cluster', 3) is not None                # ... which is looked at again after three calls
    assert calls_until('flat', 'dense', 2) is not None                  # ... and found as deep as before (the next look: six calls on)
    assert calls_until('peaked', 'cluster', 6) is not None              # peaked rows again: back on clusters with that look
    assert calls_until('peaked', 'cluster', 1) == 1
    torbi_amd.reset_path_state()


@pytest.mark.parametrize('S', [2, 3, 5, 17, 31, 32, 33, 40, 63, 64])
def test_value_only_form_of_the_wavefront_kernel(S, monkeypatch):
    """small::decode_value_kernel (csrc/small_states.hpp): no backpointers, the posterior rows kept and the first argmax
    recomputed along the decoded path from the matrix in the LDS -- AUTO's choice from 32 padded states and 512 sequences
    up, forced here for every state count: ties on a coarse grid, -inf entries, a state nobody can come from, ragged
    lengths incl. 1, more sequences than one workgroup holds; and the byte-backpointer form forced on the same inputs."""
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(S)
    for B, T in [(1, 1), (5, 5), (70, 67), (6, 130), (530, 19)]:
        obs = -rng.integers(0, 6, size=(B, T, S)).astype(np.float32)
        trans = -rng.integers(0, 5, size=(S, S)).astype(np.float32)
        init = -rng.integers(0, 3, size=(S,)).astype(np.float32)
        trans[rng.random((S, S)) < 0.2] = -np.inf
        obs[rng.random((B, T, S)) < 0.05] = -np.inf
        if S > 2:
            trans[:, S - 1] = -np.inf
        frames = np.resize(np.array([T, 1, max(T - 1, 1), max(T - 3, 1), max(T // 2, 1), max(T - 16, 1), max(T - 17, 1)],
                                    np.int32), B).astype(np.int32)
        want, post = oracle.decode(obs, frames, trans, init, return_posterior=True)
        args = [torch.tensor(x, device=dev) for x in (obs, frames, trans, init)]
        ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
        for form, name in (('1', 'small::decode_value_kernel<'), ('0', 'small::decode_kernel<')):
            monkeypatch.setenv('TORBI_HIP_SMALL_VALUE', form)
            got = torbi_amd.decode(*args, workspace=ws)
            assert viterbi.last_forward_kernel().startswith(name)
            np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f'{B} x {T} x {S} form {form}')
            last = viterbi.read_posterior(ws, args[1], B, T, S).cpu().numpy()
            assert np.array_equal(last.view(np.uint32), post.view(np.uint32))
    monkeypatch.delenv('TORBI_HIP_SMALL_VALUE')


@pytest.mark.parametrize('path', ['cluster', 'resident', 'band'])
@pytest.mark.parametrize('segments', ['1', '3', '8', '16'])
def test_backtrace_in_speculative_segments_is_the_whole_path(path, segments, monkeypatch):
    """Behind a time-resident or band forward launch, one batch of few sequences is walked back in K segments per sequence,
    each from the first argmax of a posterior row, and joined from the end of the path (csrc/lazy_backtrace.hpp,
    chase_segment / stitch_segments).  Asserted for K = 1 (whole paths), 3, 8, 16: the oracle's indices on ragged lengths
    that include 1, 2, fewer steps than segments and the full length; on a coarse grid of values (every joint a tie);
    with rows of -inf; and the counters of the joints in torbi_hip_scan_stats [122], [123]."""
    monkeypatch.setenv('TORBI_HIP_BACKTRACE_SEGMENTS', segments)
    dev = torch.device('cuda:0')
    B, T, S = 40, 61, 360
    for variant in ('plain', 'ties', 'inf'):
        obs, trans, init = synth.problem(B, T, S, seed=11)
        if path == 'band':
            trans = _banded(S, 9, 14, seed=5)
        if variant == 'ties':
            obs, trans, init = (np.where(np.isfinite(x), np.round(x * 2.0) / 2.0, x).astype(np.float32) for x in (obs, trans, init))
        if variant == 'inf':
            obs = obs.copy()
            obs[:, 20:23, ::2] = -np.inf
            obs[3, 30, :] = -np.inf
        frames = np.full((B,), T, np.int32)
        frames[:12] = [1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 33, 60]
        want = oracle.decode(obs, frames, trans, init)
        args = [torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs, frames, trans, init)]
        space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
        prof = []
        got = torbi_amd.decode(*args, path=path, workspace=space, _profile=prof)
        assert viterbi.ROUTES[int(prof[3])] == path
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f'{path} {segments} {variant}')
        stats = viterbi.scan_stats(space, B, T, S).cpu()
        if segments == '1':
            assert int(stats[122]) == 0 and int(stats[123]) == 0
        else:
            assert int(stats[122]) == int((np.clip(frames, 1, T) - 1).sum())
            assert int(stats[123]) <= int(stats[122])


def test_cluster_exchange_survives_the_one_nan_it_uses_as_absent(monkeypatch):
    """The cluster form's exchange marks a slice that has not arrived with one NaN bit pattern (resident_forward.hpp,
    kAbsentBits).  NaNs are out of contract -- but an observation that carries exactly that pattern must not hang the
    call: the members that wait for the poisoned row give up within the wait budget, the tile is decoded again by the
    repair launch, every other sequence still equals the oracle's."""
    import time
    monkeypatch.setenv('TORBI_HIP_CLUSTER_WAIT_US', '3000')
    dev = torch.device('cuda:0')
    B, T, S = 48, 12, 1440
    obs, trans, init = synth.problem(B, T, S, seed=21)
    frames = np.full((B,), T, np.int32)
    want = oracle.decode(obs, frames, trans, init)
    poisoned = obs.copy()
    poisoned.view(np.uint32)[5, 6, 700] = 0x7fd5a5a5            # item 5 (tile 0), timestep 6
    args = [torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (poisoned, frames, trans, init)]
    space = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = torbi_amd.decode(*args, path='cluster', workspace=space).cpu().numpy()
    assert time.perf_counter() - t0 < 5.0
    stats = viterbi.scan_stats(space, B, T, S).cpu()
    assert int(stats[127]) > 0                                  # somebody gave up, the repair launch ran
    clean = [b for b in range(B) if b != 5]
    np.testing.assert_array_equal(got[clean], want[clean])
