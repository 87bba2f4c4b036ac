"""Would exact early termination INSIDE the band pay?  A CPU simulation of the whole-tile band kernel's scan
(csrc/band_tile_forward.hpp) under lock step, on the bench's rows and on rows a pitch tracker would produce.

The kernel evaluates every finite cell: a wave walks the 44 dquads (4 diagonals each) of a 64-next-state block for 16 items.
An exact bound could end a walk early: every candidate not yet examined is at most
    fl( max of the window rows still to come  +  largest band entry still to come )
(rounding is monotone: the bound of csrc/pruned_forward.hpp:8-13, values only, ties need no care); once that is <= the running
best of EVERY one of the wave's 64 x 16 outputs the rest of the walk cannot change a value.  The wave is in lock step: one
output that is not final keeps all 1 024 walking.

Orders simulated, dquads walked per wave and timestep (of Dq = 44):
    ltr        left to right as today; window bound = the item's suffix maximum (one scan per row and item)
    ltr-exact  ... bound = the maximum over exactly the rows the lane still reads (what no cheaper bound can beat)
    ltr-oracle ... stop when every output HAS its final value (no bound at all: the floor of any left-to-right scheme)
    ltr-16     what the kernel's LDS has room for: suffix maxima per granule of 16 rows, a look every 4 dquads;
               busiest-SIMD: the timestep this gives (12 waves x 2 blocks on 4 SIMDs, the busiest SIMD's sum) -- before the
               cost of the suffix pass (~4 %) and of the looks (~5 %)
    out        centre -> right edge, then centre -> left edge (two monotone walks, each with its own stop),
               suffix / prefix maxima
    out-exact, out-oracle   as above

    python tools/band_bound_sim.py [items] [frames]
"""
import math
import sys

import numpy as np

S, HALF = 1440, 87.2
HL = HR = 87
DQ = (HL + HR + 1 + 3) // 4


def pitch_band():
    x = np.arange(S)
    tri = np.clip(HALF - np.abs(x[:, None] - x[None, :]), 0, None).astype(np.float64)
    tri /= tri.sum(axis=1, keepdims=True)
    with np.errstate(divide='ignore'):
        return np.log(tri).astype(np.float32)


def rows(kind, B, T, rng):
    logits = rng.standard_normal((B, T, S)).astype(np.float32) * 2.0
    if kind == 'peaked':                    # bench.py: a random centre per frame
        centre = rng.integers(0, S, (B, T, 1))
    else:                                   # a centre that wanders 12 bins a frame (a pitch track)
        centre = (S / 2 + np.cumsum(rng.standard_normal((B, T, 1)) * 12.0, axis=1)) % S
    logits -= (np.abs(np.arange(S)[None, None, :] - centre) / 12.0) ** 2
    if kind == 'mixed':                     # stretches of 50 frames without a peak (unvoiced), 50 with one
        flat = ((np.arange(T) // 50) % 2 == 1)[None, :, None]
        logits = np.where(flat, rng.standard_normal((B, T, S)).astype(np.float32) * 0.3, logits)
    m = logits.max(axis=-1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(axis=-1, keepdims=True))
    return np.maximum(logits - lse, math.log(np.finfo(np.float32).tiny)).astype(np.float32)


def diagonals(trans):
    """D[j][d] = trans[j][j + d - HL] (-inf outside the matrix), d = 0 .. 4 DQ - 1."""
    D = np.full((S, 4 * DQ), -np.inf, np.float32)
    for d in range(HL + HR + 1):
        j = np.arange(S)
        i = j + d - HL
        ok = (i >= 0) & (i < S)
        D[j[ok], d] = trans[j[ok], i[ok]]
    return D


def simulate(kind, B, T, seed=7):
    rng = np.random.default_rng(seed)
    obs = rows(kind, B, T, rng)
    D = diagonals(pitch_band())
    init = np.full((S,), math.log(1.0 / S), np.float32)
    post = obs[:, 0, :] + init[None, :]
    pad = 4 * DQ
    nblk = (S // 4 + 15) // 16
    walked = {k: [] for k in ('ltr', 'ltr-exact', 'ltr-oracle', 'out', 'out-exact', 'out-oracle', 'ltr-16')}
    simd_full, simd_walk = [], []       # 'ltr-16' per tile and timestep: dquads of the busiest SIMD, all / walked
    cq = DQ // 2                    # first dquad of the walk to the right
    for t in range(1, T):
        # window of the previous row, padded: W[b][w] = post[b][w - HL]; outside the matrix -inf (never wins, never bounds)
        W = np.full((B, S + pad + 8), -np.inf, np.float32)
        W[:, HL:HL + S] = post
        suf = np.maximum.accumulate(W[:, ::-1], axis=1)[:, ::-1]          # max of rows >= w
        # what fits the kernel's LDS beside window and rings (5.8 KB): the suffix maximum per GRANULE of 16 rows (from the
        # granule's first row on), looked at every 4 dquads
        suf16 = suf[:, (np.arange(W.shape[1]) // 16) * 16]
        per_tile = np.zeros((B // 16, nblk), np.int64)
        pre = np.maximum.accumulate(W, axis=1)                           # max of rows <= w
        new = np.empty_like(post)
        for blk in range(nblk):
            j0 = 64 * blk
            nj = min(64, S - j0)
            j = j0 + np.arange(nj)
            # cand[b][jj][d] = W[b][j + d] + D[j][d]
            idx = j[:, None] + np.arange(4 * DQ)[None, :]
            cand = W[:, idx] + D[j][None, :, :]                            # (B, nj, 4 DQ) float32 adds
            per_q = cand.reshape(B, nj, DQ, 4).max(axis=3)                # best of every dquad
            final = per_q.max(axis=2)
            new[:, j0:j0 + nj] = final
            tmax_q = D[j].reshape(nj, DQ, 4).max(axis=2).max(axis=0)      # largest band entry of the block per dquad
            jg4 = (j // 4) * 4                                            # first next-state of the lane's group
            for tile in range(B // 16):
                b = slice(16 * tile, 16 * tile + 16)
                pq, fin = per_q[b], final[b]
                # ---- left to right
                run = np.maximum.accumulate(pq, axis=2)                   # best after dquad q
                rem_t = np.concatenate([np.maximum.accumulate(tmax_q[::-1])[::-1][1:], [-np.inf]])    # band max of dquads > q
                first_row = jg4[:, None] + 4 * (np.arange(DQ)[None, :] + 1)                            # window row of the first cell to come
                last_row = jg4 + 3 + HL + HR                                                           # ... of the last
                b_suf = suf[b][:, first_row] + rem_t[None, None, :]
                done = (b_suf <= run).all(axis=(0, 1))
                walked['ltr'].append(int(np.argmax(done)) + 1 if done.any() else DQ)
                looks = np.arange(3, DQ, 4)                               # behind dquads 3, 7, ...
                d16 = (suf16[b][:, first_row[:, looks]] + rem_t[None, None, looks] <= run[:, :, looks]).all(axis=(0, 1))
                n16 = int(looks[int(np.argmax(d16))]) + 1 if d16.any() else DQ
                walked['ltr-16'].append(n16)
                per_tile[tile, blk] = n16
                # exact window maximum of the rows still to come
                ex = np.full((16, nj, DQ), -np.inf, np.float32)
                for q in range(DQ - 1):
                    lo = jg4 + 4 * (q + 1)
                    span = np.arange(0, 4 * (DQ - q - 1) + 3)
                    ex[:, :, q] = W[b][:, (lo[:, None] + span[None, :])].max(axis=2)
                done = (ex + rem_t[None, None, :] <= run).all(axis=(0, 1))
                walked['ltr-exact'].append(int(np.argmax(done)) + 1 if done.any() else DQ)
                done = (run == fin[:, :, None]).all(axis=(0, 1))
                walked['ltr-oracle'].append(int(np.argmax(done)) + 1)
                # ---- centre out: right walk cq .. DQ - 1, then left walk cq - 1 .. 0
                right = np.maximum.accumulate(pq[:, :, cq:], axis=2)
                rem_r = np.concatenate([np.maximum.accumulate(tmax_q[cq:][::-1])[::-1][1:], [-np.inf]])
                fr = jg4[:, None] + 4 * (cq + np.arange(DQ - cq)[None, :] + 1)
                steps = {}
                for name, bound in (('out', suf[b][:, fr]), ('out-exact', ex[:, :, cq:])):
                    d_r = (bound + rem_r[None, None, :] <= right).all(axis=(0, 1))
                    n_r = int(np.argmax(d_r)) + 1 if d_r.any() else DQ - cq
                    best_r = right[:, :, n_r - 1] if True else None
                    # what the right walk skipped is provably <= best_r; the left walk starts from best_r
                    left = np.maximum(np.maximum.accumulate(pq[:, :, :cq][:, :, ::-1], axis=2), best_r[:, :, None])
                    tl = tmax_q[:cq][::-1]
                    rem_l = np.concatenate([np.maximum.accumulate(tl[::-1])[::-1][1:], [-np.inf]])
                    # rows still to come on the left after walking dquads cq-1 .. cq-1-k: rows <= jg4 + 3 + 4 (cq - 1 - k) - 1 + 3
                    lr = jg4[:, None] + 4 * (cq - 1 - np.arange(cq)[None, :]) + 2
                    if name == 'out':
                        bl = pre[b][:, lr]
                    else:
                        bl = np.full((16, nj, cq), -np.inf, np.float32)
                        for k in range(cq - 1):
                            hi = jg4 + 4 * (cq - 1 - k) + 2
                            span = np.arange(0, 4 * (cq - 1 - k) + 3)
                            bl[:, :, k] = W[b][:, np.maximum(hi[:, None] - span[None, :], 0)].max(axis=2)
                    d_l = (bl + rem_l[None, None, :] <= left).all(axis=(0, 1))
                    n_l = int(np.argmax(d_l)) + 1 if d_l.any() else cq
                    steps[name] = n_r + n_l
                walked['out'].append(steps['out'])
                walked['out-exact'].append(steps['out-exact'])
                d_r = (np.maximum.accumulate(pq[:, :, cq:], axis=2) == pq[:, :, cq:].max(axis=2)[:, :, None]).all(axis=(0, 1))
                n_r = int(np.argmax(d_r)) + 1
                need_left = (pq[:, :, :cq].max(axis=2) > pq[:, :, cq:].max(axis=2))
                if need_left.any():
                    lrun = np.maximum.accumulate(pq[:, :, :cq][:, :, ::-1], axis=2)
                    reach = np.where(need_left[:, :, None], lrun == fin[:, :, None], True).all(axis=(0, 1))
                    n_l = int(np.argmax(reach)) + 1
                else:
                    n_l = 0
                walked['out-oracle'].append(n_r + n_l)
        # the kernel's timestep: 12 waves x 2 blocks, wave w on SIMD w % 4 -- the busiest SIMD's dquads
        simd_of = (np.arange(nblk) // 2) % 4
        for tile in range(B // 16):
            simd_walk.append(max(int(per_tile[tile, simd_of == s_].sum()) for s_ in range(4)))
            simd_full.append(max(int((simd_of == s_).sum()) * DQ for s_ in range(4)))
        post = obs[:, t, :] + new
    out = {k: float(np.mean(v)) for k, v in walked.items()}
    out['busiest-SIMD'] = DQ * float(np.sum(simd_walk)) / float(np.sum(simd_full))      # (as dquads of 44, for the same print)
    return out


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    print(f'{B} items x {T} frames x {S} states, pitch band (reach {HL}), Dq = {DQ} dquads per wave and timestep; '
          f'mean dquads walked per wave (16 items x 64 next-states in lock step)')
    for kind in ('peaked', 'smooth', 'mixed'):
        r = simulate(kind, B, 110 if kind == 'mixed' and T < 110 else T)
        print(f'{kind:7s} ' + '  '.join(f'{k} {v:5.1f} ({DQ / v:4.2f}x)' for k, v in r.items()), flush=True)
