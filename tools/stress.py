"""Randomised parity stress (GPU box): random shapes, lengths, -inf densities, tie levels and peaked rows under the six
named forward paths (and the CPU twin) against the C oracle.   python tools/stress.py [cases] [seed]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, torbi_amd
from torbi_amd import synth, viterbi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
bad = 0
uniform_cases = 0
t0 = time.time()
for c in range(cases):
    S = int(rng.choice([rng.integers(1, 80), rng.integers(2, 258), rng.integers(16, 560) * 4, rng.integers(64, 2200),
                        rng.integers(513, 1025) * 4]))
    B = int(rng.choice([rng.integers(1, 20), rng.integers(17, 70), rng.integers(60, 160), rng.integers(250, 300)]))
    T = int(rng.integers(1, 10)) if B > 16 else int(rng.integers(1, 60))     # a handful of sequences: many timesteps
    if S <= 256:                                                             # one launch per decode: long walks back
        T = int(rng.integers(1, 400))
    if B * T * S * S > 6e9:
        B = max(1, int(6e9 / (T * S * S)))
    obs, trans, init = synth.problem(B, T, S, seed=int(rng.integers(1 << 30)))
    kind = rng.integers(10)
    background = None
    if kind == 1:      # heavy ties
        obs = np.round(obs / 4) * 4; trans = np.round(trans / 8) * 8
    elif kind == 2:    # random -inf entries
        trans = np.where(rng.random((S, S)) < rng.choice([0.3, 0.9, 0.99]), -np.inf, trans).astype(np.float32)
    elif kind == 3:    # band
        trans = np.where(np.abs(np.arange(S)[:, None] - np.arange(S)[None, :]) > rng.integers(1, max(2, S // 4)), -np.inf, trans).astype(np.float32)
    elif kind == 4:    # -inf observations / initial
        obs = np.where(rng.random(obs.shape) < 0.2, -np.inf, obs).astype(np.float32)
        init = np.where(rng.random(S) < 0.5, -np.inf, init).astype(np.float32)
    elif kind == 5:    # tiny spread: nothing prunable
        trans = (trans * np.float32(2.0 ** -int(rng.integers(8, 20)))).astype(np.float32)
    elif kind == 6:    # peaked rows (posteriorgram-like: a narrow peak per frame over noise), dense matrix
        centre = rng.integers(0, S, size=(B, T, 1))
        width = float(rng.choice([3.0, 12.0, 40.0]))
        obs = (obs / 4 - ((np.abs(np.arange(S)[None, None, :] - centre)) / width) ** 2).astype(np.float32)
    elif kind == 7:    # NaN / +inf somewhere (csrc/nonfinite.hpp: the reference's results on every route)
        for _ in range(int(rng.integers(1, 4))):
            where = int(rng.integers(4))
            bad_value = np.float32(rng.choice([np.nan, np.inf]))
            if where == 0: obs[rng.integers(B), rng.integers(T), rng.integers(S)] = bad_value
            elif where == 1: trans[rng.integers(S), rng.integers(min(S, 3))] = bad_value
            elif where == 2: init[rng.integers(min(S, 2))] = bad_value
            else: obs[rng.integers(B), rng.integers(T), 0] = bad_value
        trans = np.where(rng.random((S, S)) < 0.05, -np.inf, trans).astype(np.float32)
    elif kind in (8, 9):   # a band with ONE constant outside it (band_tile_forward.hpp), from "never matters" to "wins everywhere"
        reach = int(rng.integers(1, max(2, min(S // 4, 120))))
        background = np.float32(rng.choice([-87.33654, -20.0, -3.0, 1.5]))
        trans = np.where(np.abs(np.arange(S)[:, None] - np.arange(S)[None, :]) > reach, background, trans).astype(np.float32)
        if kind == 9:
            centre = rng.integers(0, S, size=(B, T, 1))
            obs = (obs / 4 - np.abs(np.arange(S)[None, None, :] - centre) * 1.5).clip(min=-87.0).astype(np.float32)
    frames = rng.integers(1, T + 1, size=B).astype(np.int32)
    want = oracle.decode(obs.astype(np.float32), frames, trans.astype(np.float32), init.astype(np.float32), num_threads=oracle.max_threads())
    args = [torch.as_tensor(np.ascontiguousarray(x)).to(dev) for x in (obs.astype(np.float32), frames, trans.astype(np.float32), init.astype(np.float32))]
    twin = torbi_amd.decode_cpu(*[torch.as_tensor(np.ascontiguousarray(x)) for x in (obs.astype(np.float32), frames, trans.astype(np.float32), init.astype(np.float32))],
                                num_threads=int(rng.integers(1, 9))).numpy()
    if not np.array_equal(twin, want):
        bad += 1
        print('MISMATCH', dict(B=B, T=T, S=S, kind=int(kind), path='cpu twin'), int((twin != want).sum()))
    for path in ('auto', 'dense', 'pruned', 'resident', 'cluster', 'held') + (('band',) if kind in (3, 8, 9) else ()):
        viterbi.set_forward_path('auto' if path == 'band' else path)
        if path == 'band':      # both forms of the band kernel on any number of items
            os.environ['TORBI_HIP_BAND_FORM'] = 'tile' if c % 2 else 'split'
        got = torbi_amd.decode(*args).cpu().numpy()
        os.environ.pop('TORBI_HIP_BAND_FORM', None)
        if not np.array_equal(got, want):
            bad += 1
            print('MISMATCH', dict(B=B, T=T, S=S, kind=int(kind), path=path, used=viterbi.forward_path(B, S)), int((got != want).sum()))
    # the uniform-transition entry on the same observations (every other case whose state count it takes): scores as they
    # are, and as probabilities through the fused log + epsilon round trip, against the oracle on the materialised matrix
    if S % 4 == 0 and S <= 4096 and c % 2 == 0:
        viterbi.set_forward_path('auto')
        cu = np.float32(math.log(1.0 / S))
        full = np.full((S, S), cu, np.float32)
        want_u = oracle.decode(obs.astype(np.float32), frames, full, init.astype(np.float32), num_threads=oracle.max_threads())
        got_u = torbi_amd.decode_uniform(args[0], args[1], float(cu), args[3]).cpu().numpy()
        uniform_cases += 1
        if not np.array_equal(got_u, want_u):
            bad += 1
            print('MISMATCH', dict(B=B, T=T, S=S, kind=int(kind), path='uniform'), int((got_u != want_u).sum()))
        if kind != 4:                                # (a row of -inf scores has no softmax)
            probs = torch.softmax(args[0], dim=-1)
            scores = torch.log(probs)
            scores.exp_()
            scores += torch.finfo(torch.float32).tiny
            scores.log_()
            want_p = oracle.decode(scores.cpu().numpy(), frames, full, init.astype(np.float32), num_threads=oracle.max_threads())
            got_p = torbi_amd.decode_uniform(probs, args[1], float(cu), args[3], probabilities=True).cpu().numpy()
            if not np.array_equal(got_p, want_p):
                bad += 1
                print('MISMATCH', dict(B=B, T=T, S=S, kind=int(kind), path='uniform, probabilities'), int((got_p != want_p).sum()))
viterbi.set_forward_path('auto')
print(f'{cases} cases x (6 HIP paths + the CPU twin) + {uniform_cases} x the uniform entry (scores, probabilities), {bad} mismatches, '
      f'{time.time() - t0:.0f} s')
sys.exit(1 if bad else 0)
