"""LDS bank conflicts of the list blocks: how far is the greedy arrangement (pruned_forward.hpp arrange_blocks_kernel) from
what a block allows?  4 rows x 16 entries, residues mod 4 uniform; position 0 is fixed in every row; a position costs as
many LDS cycles as its most frequent residue.  Compares: as sorted, the greedy pass, greedy + pairwise improvement
(swap two entries of one row when that lowers the cost), and a (weak) lower bound from the residue totals.
    python tools/arrange_sim.py"""
import numpy as np
rng = np.random.default_rng(0)

def cost(block):                       # block[row][pos] residues
    return sum(np.bincount(block[:, p], minlength=4).max() for p in range(block.shape[1]))

def greedy(block):
    out = block.copy()
    present = np.zeros((16, 4), int)
    for p in range(16):
        present[p, block[0, p]] += 1
    for r in range(1, 4):
        ent = list(block[r])
        remaining = np.bincount(ent[1:], minlength=4)
        present[0, ent[0]] += 1
        used = [False] * 16
        used[0] = True
        for p in range(1, 16):
            best, key = -1, 1 << 30
            for e in range(1, 16):
                if used[e]:
                    continue
                k = present[p, ent[e]] * 64 - remaining[ent[e]]
                if k < key:
                    key, best = k, e
            used[best] = True
            remaining[ent[best]] -= 1
            present[p, ent[best]] += 1
            out[r, p] = ent[best]
    return out

def improve(block):
    out = block.copy()
    changed = True
    while changed:
        changed = False
        for r in range(4):
            for a in range(1, 16):
                for b in range(a + 1, 16):
                    if out[r, a] == out[r, b]:
                        continue
                    before = np.bincount(out[:, a], minlength=4).max() + np.bincount(out[:, b], minlength=4).max()
                    out[r, a], out[r, b] = out[r, b], out[r, a]
                    after = np.bincount(out[:, a], minlength=4).max() + np.bincount(out[:, b], minlength=4).max()
                    if after < before:
                        changed = True
                    else:
                        out[r, a], out[r, b] = out[r, b], out[r, a]
    return out

n = 400
tot = np.zeros(4)
for _ in range(n):
    block = rng.integers(0, 4, size=(4, 16))
    g = greedy(block)
    totals = np.bincount(block[:, 1:].ravel(), minlength=4)
    # a residue that occurs n times in the 15 free positions doubles up in at least n - 15 of them
    bound = np.bincount(block[:, 0], minlength=4).max() + 15 + max(0, np.maximum(totals - 15, 0).max())
    tot += [cost(block), cost(g), cost(improve(g)), bound]
print('LDS cycles per position (1.0 = no conflict): as sorted %.3f, greedy %.3f, greedy + swaps %.3f, lower bound %.3f' % tuple(tot / n / 16))
