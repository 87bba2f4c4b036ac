"""Generate the committed golden vectors in tests/golden/ from the REFERENCE operator.

Run in the build container (where /root/reference exists):

    python tests/golden/generate.py

The outputs (`*.npz`) are data: inputs and the index sequences the reference's own
compiled CPU operator (oracle/_ref, built by oracle/build.py from
/root/reference/torbi/csrc/{ops,viterbi}.cpp) returned for them.  Large inputs are not
stored; they are regenerated bit-identically by torbi_amd/synth.py from (shape, seed).

Cases follow SURVEY.md section 8(c):
  G0 toy (reference tests/test_core.py:9-25 and README) incl. defaults and batch_frames
  G1 orientation (non-symmetric transition, ragged batch_frames incl. 1)
  G2 ties (all-equal inputs; quantised inputs with ties across lane/chunk boundaries;
     final-state ties)
  G3 -inf transitions (banded, diagonal)
  G4 S=1440 T=500: B=1 and the first 4 items of the headline batch
  G5 S=4096 T=2000: first 2 items of the large-S config
  GX shape sweep (S in 1..257, T in 1..33, ragged frames)
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from torbi_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
TINY = float(torch.finfo(torch.float32).tiny)


def ref(obs, frames, trans, init, threads=8):
    return oracle.ref_decode(obs, frames, trans, init, num_threads=threads).numpy()


def api_preprocess(observation, transition=None, initial=None, log_probs=False):
    """What reference from_probabilities does before `decode` on CPU (torbi/core.py:146-197)."""
    import math
    observation = torch.as_tensor(observation)
    B, T, S = observation.shape
    if initial is None:
        initial = torch.full((S,), math.log((1. / S) + TINY), dtype=torch.float32)
    else:
        initial = torch.as_tensor(initial)
        if not log_probs:
            initial = torch.log(initial)
    if transition is None:
        transition = torch.full((S, S), math.log(1. / S), dtype=torch.float32)
    else:
        transition = torch.as_tensor(transition)
        if not log_probs:
            transition = torch.log(transition)
    if not log_probs:
        observation = torch.log(observation)
    observation = observation.to(dtype=torch.float32).clone()
    torch.exp_(observation)
    observation += TINY
    torch.log_(observation)
    return observation.numpy(), transition.numpy(), initial.numpy()


def main():
    small = {}
    names = []

    def add(name, obs, frames, trans, init):
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        frames = np.asarray(frames, dtype=np.int32)
        trans = np.ascontiguousarray(trans, dtype=np.float32)
        init = np.ascontiguousarray(init, dtype=np.float32)
        idx = ref(obs, frames, trans, init)
        assert (idx == ref(obs, frames, trans, init, threads=1)).all()
        small[name + '/observation'] = obs
        small[name + '/batch_frames'] = frames
        small[name + '/transition'] = trans
        small[name + '/initial'] = init
        small[name + '/indices'] = idx.astype(np.int32)
        names.append(name)
        print(name, obs.shape, idx[0, :8])

    # ---- G0 toy --------------------------------------------------------------
    toy_obs = np.array([[[0.25, 0.5, 0.25], [0.25, 0.25, 0.5], [0.33, 0.33, 0.33]]], np.float32)
    toy_tr = np.array([[0.5, 0.25, 0.25], [0.33, 0.34, 0.33], [0.25, 0.25, 0.5]], np.float32)
    toy_in = np.array([0.4, 0.35, 0.25], np.float32)
    o, t, i = api_preprocess(toy_obs, toy_tr, toy_in)
    add('g0_toy', o, [3], t, i)
    assert small['g0_toy/indices'].tolist() == [[1, 2, 2]]
    o, t, i = api_preprocess(toy_obs)
    add('g0_toy_defaults', o, [3], t, i)
    o, t, i = api_preprocess(toy_obs, toy_tr, toy_in)
    add('g0_toy_frames2', o, [2], t, i)
    small['g0_toy/probabilities'] = toy_obs
    small['g0_toy/transition_probabilities'] = toy_tr
    small['g0_toy/initial_probabilities'] = toy_in

    # ---- G1 orientation ------------------------------------------------------
    rng = np.random.default_rng(1234)
    S, T, B = 17, 40, 3
    add('g1_orientation', rng.uniform(-8, 0, (B, T, S)), [40, 25, 1],
        rng.uniform(-8, 0, (S, S)), rng.uniform(-8, 0, (S,)))

    # ---- G2 ties ---------------------------------------------------------------
    S, T, B = 70, 9, 2
    add('g2_all_zero', np.zeros((B, T, S)), [9, 5], np.zeros((S, S)), np.zeros(S))
    S, T, B = 130, 30, 4
    add('g2_quantised', -rng.integers(0, 4, (B, T, S)).astype(np.float32), [30, 30, 17, 2],
        -rng.integers(0, 3, (S, S)).astype(np.float32), -rng.integers(0, 2, (S,)).astype(np.float32))
    # crafted: maxima at index pairs straddling 32/64 boundaries with exact ties
    S, T, B = 200, 6, 5
    obs = np.full((B, T, S), -5.0, np.float32)
    tr = np.full((S, S), -3.0, np.float32)
    init = np.full((S,), -1.0, np.float32)
    pairs = [(1, 32), (1, 33), (31, 64), (63, 65), (64, 128)]
    for b, (lo, hi) in enumerate(pairs):
        obs[b, :, lo] = -1.0
        obs[b, :, hi] = -1.0
    add('g2_crafted_ties', obs, [6, 6, 6, 4, 6], tr, init)
    S, T, B = 9, 4, 1
    obs = np.zeros((B, T, S), np.float32)
    obs[0, -1, 3] = 1.0
    obs[0, -1, 6] = 1.0
    add('g2_final_tie', obs, [4], np.zeros((S, S)), np.zeros(S))

    # ---- G3 -inf transitions -------------------------------------------------
    S, T, B = 50, 25, 3
    add('g3_banded', rng.uniform(-8, 0, (B, T, S)), [25, 25, 11],
        synth.banded_transition(S, 4), rng.uniform(-8, 0, (S,)))
    diag = np.full((S, S), -np.inf, np.float32)
    np.fill_diagonal(diag, 0.0)
    add('g3_diagonal', rng.uniform(-8, 0, (B, T, S)), [25, 3, 25], diag, rng.uniform(-8, 0, (S,)))
    inf_obs = rng.uniform(-8, 0, (2, 12, 40)).astype(np.float32)
    inf_obs[0, 3, :20] = -np.inf
    inf_obs[1, 5, :] = -np.inf
    add('g3_inf_observation', inf_obs, [12, 12], rng.uniform(-8, 0, (40, 40)), rng.uniform(-8, 0, (40,)))

    # ---- GX shape sweep --------------------------------------------------------
    sweep = [(1, 1, 1), (1, 2, 1), (2, 1, 5), (1, 5, 2), (3, 7, 3), (2, 33, 63), (2, 20, 64),
             (3, 19, 65), (5, 12, 127), (2, 9, 129), (4, 16, 257), (70, 6, 31), (130, 3, 45)]
    for n, (B, T, S) in enumerate(sweep):
        o, t, i = synth.problem(B, T, S, seed=100 + n)
        fr = np.clip(synth.lengths(B, 1, T, seed=n), 1, T)
        fr[0] = T
        add(f'gx_sweep_{B}x{T}x{S}', o, fr, t, i)

    small['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'golden_small.npz'), **small)

    # ---- large, inputs regenerated from (shape, seed) ---------------------------
    large = {}
    lnames = []

    def add_large(name, B, T, S, seed, frames=None):
        o, t, i = synth.problem(B, T, S, seed=seed)
        fr = np.full((B,), T, np.int32) if frames is None else np.asarray(frames, np.int32)
        idx = ref(o, fr, t, i).astype(np.int32)
        large[name + '/shape'] = np.array([B, T, S, seed], np.int64)
        large[name + '/batch_frames'] = fr
        large[name + '/indices'] = idx
        large[name + '/sha256'] = np.array(hashlib.sha256(idx.tobytes()).hexdigest())
        lnames.append(name)
        print(name, (B, T, S), idx[0, :8], idx[0, -8:], flush=True)

    add_large('g4_c2_1x500x1440', 1, 500, 1440, seed=0)
    add_large('g4_c3_first4_4x500x1440', 4, 500, 1440, seed=0)
    add_large('g4_ragged_6x300x1440', 6, 300, 1440, seed=5, frames=[300, 1, 2, 150, 299, 77])
    add_large('g4_mid_3x64x360', 3, 64, 360, seed=6)
    add_large('g5_c5_first2_2x2000x4096', 2, 2000, 4096, seed=0)
    large['names'] = np.array(lnames)
    np.savez_compressed(os.path.join(OUT, 'golden_large.npz'), **large)


if __name__ == '__main__':
    main()
