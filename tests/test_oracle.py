"""The CPU oracle (oracle/viterbi_oracle.c) against the reference's golden vectors.

These pin the oracle: every committed vector was produced by the reference's own compiled
CPU operator (tests/golden/generate.py), and the toy is the reference's only known-answer
test (reference tests/test_core.py:7-25).
"""
import hashlib

import numpy as np
import pytest

import oracle
from torbi_amd import synth
from conftest import SMALL_NAMES, LARGE_NAMES


@pytest.mark.parametrize('name', SMALL_NAMES)
@pytest.mark.parametrize('mode', [0, 1])
def test_oracle_matches_reference_small(golden, name, mode):
    obs, frames, trans, init, want = golden.small_case(name)
    got = oracle.decode(obs, frames, trans, init, num_threads=2, mode=mode)
    assert got.dtype == np.int32 and got.shape == want.shape
    assert np.array_equal(got, want)


def test_toy_known_answer(golden):
    """reference tests/test_core.py:9-25: probabilities in, [1, 2, 2] out."""
    g = golden.small
    obs = np.log(g['g0_toy/probabilities'])
    tiny = np.float32(np.finfo(np.float32).tiny)
    obs = np.log(np.exp(obs) + tiny).astype(np.float32)        # core.py:193-197
    trans = np.log(g['g0_toy/transition_probabilities'])
    init = np.log(g['g0_toy/initial_probabilities'])
    assert oracle.decode(obs, [3], trans, init).tolist() == [[1, 2, 2]]


@pytest.mark.parametrize('name', LARGE_NAMES)
def test_oracle_matches_reference_large(golden, name):
    obs, frames, trans, init, want = golden.large_case(name)
    got = oracle.decode(obs, frames, trans, init, num_threads=oracle.max_threads(), mode=1)
    assert np.array_equal(got, want)
    assert hashlib.sha256(got.tobytes()).hexdigest() == str(golden.large[name + '/sha256'])


def test_reference_shaped_mode_equals_fused_at_1440():
    obs, trans, init = synth.problem(1, 12, 1440, seed=3)
    a = oracle.decode(obs, [12], trans, init, num_threads=oracle.max_threads(), mode=0)
    b = oracle.decode(obs, [12], trans, init, num_threads=1, mode=1)
    assert np.array_equal(a, b)


def test_thread_count_does_not_change_results():
    obs, trans, init = synth.problem(2, 30, 200, seed=4)
    a = oracle.decode(obs, [30, 11], trans, init, num_threads=1, mode=0)
    b = oracle.decode(obs, [30, 11], trans, init, num_threads=7, mode=0)
    assert np.array_equal(a, b)


def test_oracle_rejects_bad_lengths():
    obs, trans, init = synth.problem(1, 4, 3)
    for bad in (0, 5, -1):
        with pytest.raises(ValueError):
            oracle.decode(obs, [bad], trans, init)


def test_tail_is_filled_with_final_state():
    """viterbi.cpp:218-221: every column >= frames-1 holds the final state."""
    obs, trans, init = synth.problem(1, 10, 7, seed=9)
    got = oracle.decode(obs, [4], trans, init)
    assert (got[0, 3:] == got[0, 3]).all()


def _torbi_namespace_taken():
    # torbi_amd.torch_op.register() (dispatcher tests) defines torbi::viterbi_decode in this process; the reference's
    # own library defines it too and aborts when loaded second
    import sys
    module = sys.modules.get('torbi_amd.torch_op')
    return module is not None and module._LIBRARY is not None


@pytest.mark.skipif(not oracle.ref_available(), reason='oracle/_ref not built (no /root/reference)')
@pytest.mark.parametrize('seed', range(6))
def test_oracle_equals_reference_operator_on_fresh_inputs(seed):
    if _torbi_namespace_taken():
        pytest.skip('torbi::viterbi_decode was registered by torbi_amd.torch_op in this process')
    rng = np.random.default_rng(seed)
    B, T, S = int(rng.integers(1, 5)), int(rng.integers(1, 40)), int(rng.integers(1, 300))
    obs, trans, init = synth.problem(B, T, S, seed=1000 + seed)
    if seed % 2:   # heavy ties
        obs, trans, init = np.round(obs / 4), np.round(trans / 4), np.round(init / 4)
    frames = rng.integers(1, T + 1, B).astype(np.int32)
    want = oracle.ref_decode(obs, frames, trans, init, num_threads=2).numpy()
    for mode in (0, 1):
        assert np.array_equal(oracle.decode(obs, frames, trans, init, 3, mode), want)


@pytest.mark.skipif(not oracle.ref_available(), reason='oracle/_ref not built (no /root/reference)')
@pytest.mark.parametrize('seed', range(5))
def test_oracle_follows_the_reference_operator_on_nan_inputs(seed):
    """NaN and +inf inputs (the HIP path decodes such items again as the reference does: csrc/nonfinite.hpp): the ORACLE
    restates the reference also there: a NaN candidate at prev-state 0 is never replaced (viterbi.cpp:94-100 starts its running maximum there),
    later NaN candidates never win, and the final state is ATen's argmax, which takes the first NaN of a row
    (viterbi.cpp:218).  Checked against the reference operator itself."""
    if _torbi_namespace_taken():
        pytest.skip('torbi::viterbi_decode was registered by torbi_amd.torch_op in this process')
    rng = np.random.default_rng(100 + seed)
    B, T, S = 3, 12, 40
    obs, trans, init = synth.problem(B, T, S, seed=2000 + seed)
    nan = np.float32('nan')
    if seed == 0:
        obs[0, 3, 5] = nan                      # a NaN observation in the middle
    elif seed == 1:
        trans[7, 0] = nan                       # a NaN transition INTO prev-state 0 position of a row
        trans[9, 11] = nan
    elif seed == 2:
        init[0] = nan                           # the first posterior's first entry
    elif seed == 3:
        obs[1, T - 1, 17] = nan                 # NaN in the final row: argmax takes it
        obs[2, 0, :] = nan
    else:
        obs[0, 2, 0] = np.inf                   # +inf meets -inf: NaN candidates at prev-state 0 (they stay) and 3 (they lose)
        obs[1, 4, 3] = np.inf
        trans[6, 0] = -np.inf
        trans[5, 3] = -np.inf
    frames = np.array([T, T, T - 2], np.int32)
    want = oracle.ref_decode(obs, frames, trans, init, num_threads=2).numpy()
    for mode in (0, 1):
        assert np.array_equal(oracle.decode(obs, frames, trans, init, 2, mode), want), mode
