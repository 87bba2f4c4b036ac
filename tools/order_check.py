"""The item order and tile map a time-resident decode leaves in its workspace (layout: carve_resident in torbi_hip.hip).
(GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torbi_amd import viterbi, synth
dev = torch.device('cuda:0')
B, T, S = 512, 40, 360
obs = viterbi.fill_synthetic((B, T, S), 1, device=dev)
trans = viterbi.fill_synthetic((S, S), 2, device=dev); init = viterbi.fill_synthetic((S,), 3, device=dev)
fr = synth.lengths(B, 3, T, seed=5)
frames = torch.tensor(fr, device=dev)
ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
viterbi.decode_batches([obs], [frames], trans, init, workspaces=[ws], path='resident')
torch.cuda.synchronize()
off = (B * T * S * 4 + 255) // 256 * 256
tile_map = ws[off:off + 4 * 32].view(torch.int32).cpu().numpy()          # [kMaxGroupTiles] ints, then the item order
order = ws[off + 4 * 16384:off + 4 * 16384 + 4 * B].view(torch.int32).cpu().numpy()
want = np.lexsort((np.arange(B), -fr))
print('tile map (batch << 20 | tile), longest first:', tile_map[:8], '...', tile_map[-4:])
print('order ok:', np.array_equal(order, want), order[:20], fr[order[:20]], fr[order[-5:]])
