import numpy as np
rng = np.random.default_rng(1)
S = 1440
trans = -(rng.integers(0, 1<<24, size=(S,S)).astype(np.float32) * np.float32(2**-20))
order = np.argsort(-trans, axis=1, kind='stable')
tsorted = np.take_along_axis(trans, order, axis=1)
def uniform(n): return -(rng.integers(0, 1<<24, size=(n,S)).astype(np.float32) * np.float32(2**-20))
K = 1
allb = []
for tile in range(6):
    post = uniform(16)
    post = uniform(16) + (post[:, None, :] + trans[None, :, :]).max(2)
    srt = np.argsort(-post, axis=1, kind='stable')
    rows = rng.choice(S, size=128, replace=False)
    blocks = np.zeros((128, 16), int)
    for b in range(16):
        p = post[b]; seeds = srt[b, :K]; thr = p[srt[b, K]]
        for a, j in enumerate(rows):
            best = (p[seeds] + trans[j, seeds]).max()
            cand = p[order[j]] + tsorted[j]
            run = np.maximum(np.maximum.accumulate(cand), best)
            nb = 1
            while nb * 16 < S and tsorted[j, nb * 16] + thr > run[nb * 16 - 1]:
                nb += 1
            blocks[a, b] = nb
    allb.append(blocks)
blocks = np.concatenate(allb)          # (768 rows, 16 items)
wave = blocks.reshape(-1, 16, 16)      # 16 rows x 16 items per wave pass
print('pair mean blocks', blocks.mean(), 'wave mean', wave.max(axis=(1, 2)).mean())
for C in (4, 5, 6, 7, 8, 10, 12):
    capped = np.minimum(wave.max(axis=(1, 2)), C).mean()
    left = (blocks > C)
    extra_entries = ((blocks - C).clip(min=0) * 16)[left]
    frac = left.mean()
    # phase 2 cost model: per leftover pair ceil(extra/64) wave-steps
    steps = np.ceil(extra_entries / 64).sum() / len(wave)      # wave-steps per wave-pass
    print(f'cap {C:2d}: phase-1 blocks per pass {capped:5.2f} (vs {wave.max(axis=(1,2)).mean():.2f}), pairs left {100*frac:5.2f} % = {frac*256:5.2f} per wave pass, phase-2 wave-steps per wave-pass {steps:.2f}')
