import numpy as np
rng = np.random.default_rng(0)
S = 1440
trans = -(rng.integers(0, 1<<24, size=(S,S)).astype(np.float32) * np.float32(2**-20))
order = np.argsort(-trans, axis=1, kind='stable')          # per row: prev-states by descending t
tsorted = np.take_along_axis(trans, order, axis=1)
def peaked(n):
    logits = rng.standard_normal((n, S)).astype(np.float32) * 2
    centre = rng.integers(0, S, size=(n,1))
    logits -= ((np.abs(np.arange(S)[None,:] - centre)).astype(np.float32) / 12.0) ** 2
    m = logits.max(1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(1, keepdims=True))
    return np.maximum(logits - lse, np.log(np.finfo(np.float32).tiny)).astype(np.float32)
def uniform(n):
    return -(rng.integers(0, 1<<24, size=(n,S)).astype(np.float32) * np.float32(2**-20))
def depths(post, K, rows):
    # post: (16, S) one tile; returns depth in entries for each (row, item)
    n = post.shape[0]
    srt = np.argsort(-post, axis=1, kind='stable')
    out = np.zeros((len(rows), n), dtype=np.int64)
    for a, j in enumerate(rows):
        for b in range(n):
            p = post[b]
            seeds = srt[b, :K]
            thr = p[srt[b, K]]
            best = (p[seeds] + trans[j, seeds]).max() if K else -np.inf
            cand = p[order[j]] + tsorted[j]
            run = np.maximum.accumulate(cand)
            run = np.maximum(run, best)
            # after examining k entries (0..k-1), stop if tsorted[j,k] + thr <= run[k-1]
            bound = tsorted[j, 1:] + thr
            ok = bound <= run[:-1]
            k = np.argmax(ok) + 1 if ok.any() else S
            out[a, b] = k
    return out
rows = rng.choice(S, size=64, replace=False)
for name, gen in (('uniform', uniform), ('peaked', peaked)):
    post = gen(16)
    # one recurrence step to get realistic posteriors: post' = obs + max_i(post_i + t_ji)
    nxt = gen(16) + (post[:, None, :] + trans[None, :, :]).max(2)
    for K in (0, 1, 3, 8, 16, 32):
        d = depths(nxt, K, rows)
        blocks = np.ceil(d / 16)
        wave = blocks.reshape(4, 16, 16).max(axis=(1, 2))      # 16 rows x 16 items lock-step
        print(f'{name:8s} K={K:2d}: pair mean {d.mean():7.1f} entries, row(16 items) mean {d.max(1).mean():7.1f}, wave blocks mean {wave.mean():5.1f}')
