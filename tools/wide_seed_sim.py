"""How many explicit seeds per item would make the list scan unnecessary on peaked rows with a dense matrix?

For the bench's 'peaked_dense_transition' rows (log_softmax of randn * 2 - (d / 12)^2, uniform transitions in [-16, 0]):
candidates = the M largest posteriors of an item; a (row j, item) pair still needs its list when
max_{i in top-M}(p_i + t[j, i]) < p_(M+1) + max_i t[j, i].  Prints the share of such pairs and of 256-pair wave passes that
hold at least one, per M -- and the same for the threshold rule (every state within W of the row maximum).

    python tools/wide_seed_sim.py
"""
import numpy as np

rng = np.random.default_rng(0)
S, B, STEPS = 1440, 16, 6
trans = -(rng.integers(0, 1 << 24, size=(S, S)).astype(np.float32) * np.float32(2 ** -20))     # [next][prev]
tmax = trans.max(axis=1)


def peaked(n, width=12.0):
    logits = rng.standard_normal((n, S)).astype(np.float32) * 2
    centre = rng.integers(0, S, size=(n, 1))
    logits -= ((np.abs(np.arange(S)[None, :] - centre)).astype(np.float32) / width) ** 2
    m = logits.max(1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(1, keepdims=True))
    return np.maximum(logits - lse, np.log(np.finfo(np.float32).tiny)).astype(np.float32)


def flat(n):
    return -(rng.integers(0, 1 << 24, size=(n, S)).astype(np.float32) * np.float32(2 ** -20))


for name, gen in (('peaked w=12', lambda n: peaked(n, 12.0)), ('peaked w=3', lambda n: peaked(n, 3.0)),
                  ('peaked w=40', lambda n: peaked(n, 40.0)), ('benchmark', flat)):
    post = gen(B)
    rows = {}
    for step in range(STEPS):
        full = post[:, None, :] + trans[None, :, :]            # [b][j][i]
        exact = full.max(axis=2)
        if step >= 2:
            order = np.argsort(-post, axis=1, kind='stable')
            psort = np.take_along_axis(post, order, axis=1)
            for M in (3, 8, 16, 32, 64, 128, 256):
                idx = order[:, :M]
                best = np.max(post[np.arange(B)[:, None, None], idx[:, None, :]] + trans[:, idx].transpose(1, 0, 2), axis=2)
                need = best < psort[:, M][:, None] + tmax[None, :]          # [b][j]
                waves = need.T.reshape(S // 16, 16, B).any(axis=(1, 2))       # 16 rows x 16 items per wave pass
                rows.setdefault(('M', M), []).append((need.mean(), waves.mean(), M))
            for W in (4.0, 8.0, 12.0):
                need_all, count = [], []
                for b in range(B):
                    keep = np.nonzero(post[b] >= post[b].max() - W)[0]
                    best = (post[b][None, keep] + trans[:, keep]).max(axis=1)
                    need_all.append(best < (post[b].max() - W) + tmax)
                    count.append(len(keep))
                need = np.array(need_all)
                waves = need.T.reshape(S // 16, 16, B).any(axis=(1, 2))
                rows.setdefault(('W', W), []).append((need.mean(), waves.mean(), np.mean(count)))
        post = exact + gen(B)
    print(name)
    for key, vals in rows.items():
        v = np.mean(np.array(vals), axis=0)
        print(f'   {key[0]} = {key[1]:>5}: seeds per item {v[2]:7.1f}   pairs needing the list {v[0]:9.2e}   wave passes {v[1]:7.4f}')
