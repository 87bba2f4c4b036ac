import numpy as np
rng = np.random.default_rng(0)
S = 1440
trans = -(rng.integers(0, 1<<24, size=(S,S)).astype(np.float32) * np.float32(2**-20))
order = np.argsort(-trans, axis=1, kind='stable')
tsorted = np.take_along_axis(trans, order, axis=1)
def peaked(n, width=12.0):
    logits = rng.standard_normal((n, S)).astype(np.float32) * 2
    centre = rng.integers(0, S, size=(n,1))
    logits -= ((np.abs(np.arange(S)[None,:] - centre)).astype(np.float32) / width) ** 2
    m = logits.max(1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(1, keepdims=True))
    return np.maximum(logits - lse, np.log(np.finfo(np.float32).tiny)).astype(np.float32)
def depth_pair(p, j, seeds, thr):
    best = (p[seeds] + trans[j, seeds]).max() if len(seeds) else -np.inf
    cand = p[order[j]] + tsorted[j]
    run = np.maximum(np.maximum.accumulate(cand), best)
    # block granularity: block 0 unconditional; continue while t(first of next block) + thr > best so far
    nb = 1
    while nb * 16 < S and tsorted[j, nb * 16] + thr > run[nb * 16 - 1]:
        nb += 1
    return nb
rows = rng.choice(S, size=64, replace=False)
for width in (12.0, 3.0, 40.0):
    post = peaked(16, width)
    post = peaked(16, width) + (post[:, None, :] + trans[None, :, :]).max(2)
    srt = np.argsort(-post, axis=1, kind='stable')
    for delta in (2.0, 4.0, 8.0):
        blocks = np.zeros((64, 16), int); counts = []
        for b in range(16):
            p = post[b]; pmax = p.max(); floor = pmax - delta
            seeds = np.nonzero(p >= floor)[0]
            counts.append(len(seeds))
            if len(seeds) > 32:
                seeds, thr = srt[b, :3], p[srt[b, 3]]
            else:
                thr = floor
            for a, j in enumerate(rows):
                blocks[a, b] = depth_pair(p, j, seeds, thr)
        wave = blocks.reshape(4, 16, 16).max(axis=(1, 2))
        print(f'width {width} delta {delta}: seed counts {sorted(counts)}, pair mean {blocks.mean():.1f} blocks, wave mean {wave.mean():.1f} blocks')
