"""Builder-run crop campaign (NOT part of pytest): dm_crop_nearest_f32 against the oracle's restatement of
generate_crop_grid + grid_sample(nearest, align_corners=True) of the padded image (utils.py:571-652;
oracle.crop_nearest, pinned to the reference's fixtures g9 / g9b), on seeded random maps, crop sizes (multiples of four and not), centres on and off the map
(no exact .5 ties) and fills.      python tests/campaigns/crop_campaign.py SEED0 N"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import functional as F
from oracle import oracle

def main():
  seed0, n = int(sys.argv[1]), int(sys.argv[2])
  bad = vec = 0
  for i in range(n):
    rng = np.random.default_rng(seed0 + i)
    b, c = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    h, w = int(rng.integers(5, 400)), int(rng.integers(5, 400))
    ch = int(rng.integers(1, 300))
    cw = int(rng.integers(1, 75)) * 4 if rng.uniform() < 0.7 else int(rng.integers(1, 300))
    vec += cw % 4 == 0
    img = rng.normal(size=(b, c, h, w)).astype(np.float32)
    msk = rng.uniform(size=(b, c, h, w)) > 0.5
    centers = np.stack([rng.uniform(-20, w + 20, size=b), rng.uniform(-20, h + 20, size=b)], axis=-1)
    centers = (np.floor(centers) + rng.choice([0.0, 0.25, 0.3, 0.75], size=(b, 2))).astype(np.float32)
    fill = [None, -np.inf, 0.5, np.inf][int(rng.integers(0, 4))]
    want = oracle.crop_nearest(img, centers, cw, ch, fill)
    want_m = oracle.crop_nearest(msk, centers, cw, ch, False)
    got, got_m = F.crop_nearest(torch.from_numpy(img).cuda(), torch.from_numpy(centers), cw, ch, fill_value=fill,
                                mask=torch.from_numpy(msk).cuda())
    ok = np.array_equal(got.cpu().numpy(), want) and (fill is None or np.array_equal(got_m.cpu().numpy(), want_m))
    if not ok:
      bad += 1
      print(f"MISMATCH seed {seed0 + i}: b={b} c={c} {h}x{w} -> {ch}x{cw} fill={fill}", flush=True)
  print(f"done: {n} crop configurations, {bad} with mismatches ({vec} through the four-cells-per-thread kernel)", flush=True)

if __name__ == "__main__":
  main()
