"""cfg5's crop leg alone: sixteen 2048x2048 maps + masks -> 1024x1024 crops around fractional centres
(dm_crop_nearest_f32), four map sets in rotation (HBM-served), calls back to back between one pair of events."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import functional as F
B, M, C = 16, 2048, 1024
g = torch.Generator().manual_seed(7)
sets = []
for _ in range(4):
  top = torch.empty(B, 1, M, M).uniform_(-1, 3, generator=g).cuda()
  sets.append((top, top > 0.5))
ctr = (torch.empty(B, 2).uniform_(700, 1300, generator=g)).cuda()
for j in range(8):
  out = F.crop_nearest(sets[j % 4][0], ctr, C, C, fill_value=-np.inf, mask=sets[j % 4][1])
torch.cuda.synchronize()
res = []
for rep in range(3):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for j in range(32):
    out = F.crop_nearest(sets[j % 4][0], ctr, C, C, fill_value=-np.inf, mask=sets[j % 4][1])
  e1.record(); torch.cuda.synchronize()
  res.append(e0.elapsed_time(e1) * 1e3 / 32)
out = F.crop_nearest(sets[0][0], ctr, C, C, fill_value=-np.inf, mask=sets[0][1])
alg = B * C * C * 5 * 2
print("%s: %s us/call (%.2f of 8 TB/s at the best)  checksum %.4f mask %d" % (
    os.environ.get("DUNGEON_MAPS_AMD_LIB", "default").split("/")[-1], ["%.1f" % r for r in res],
    alg / (min(res) * 1e-6) / 8e12, float(out[0].double().sum()), int(out[1].sum())))
