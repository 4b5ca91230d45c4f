"""Builder-run fuse campaign (NOT part of pytest): fuse_topdown_maps (dm_fuse_bbox_multi_f32 -> dm_fuse_bbox_read_i32 ->
dm_fuse_scatter_multi_f32 -> dm_mask_from_map_f32) against the oracle's restatement of maps.py:2039-2287
(oracle.fuse_topdown_maps, pinned to the reference-run MapBuilder sequences g5 / g5b), on seeded hand-made source maps:
one to five maps of random size (1 .. 700 cells a side), batch 1-3, local or global, flipped or not, height or value
maps, random validity, poses, resolutions and offsets; max or min.      python tests/campaigns/fuse_campaign.py SEED0 N"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import maps as M
from oracle import oracle

f32 = lambda v: np.asarray(torch.as_tensor(v).cpu().numpy() if torch.is_tensor(v) else v, dtype=np.float32).reshape(-1)

def source_of(tm):
  p = tm.proj
  top = tm.topdown_map.cpu().numpy()
  height = np.ascontiguousarray(np.broadcast_to(tm.height_map.cpu().numpy(), top.shape))
  return dict(height=height, mask=tm.mask.cpu().numpy(), value=None if tm.is_height_map else top,
              woff=f32(p.width_offset), hoff=f32(p.height_offset), map_res=float(p.map_res), flip_h=bool(p.flip_h),
              to_global=bool(p.to_global), cam_pose=f32(p.cam_pose).reshape(-1, 3))

def main():
  seed0, n = int(sys.argv[1]), int(sys.argv[2])
  bad = empty = big = 0
  for i in range(n):
    rng = np.random.default_rng(seed0 + i)
    b = int(rng.integers(1, 4))
    semantic = bool(rng.integers(2))
    c = int(rng.integers(1, 4)) if semantic else 1
    red = "max" if rng.integers(3) else "min"
    fill = float(rng.choice([0.0, 0.5])) if semantic else (-np.inf if red == "max" else np.inf)
    tms = []
    for k in range(int(rng.integers(1, 6))):
      h = int(rng.integers(1, 700)) if rng.uniform() < 0.3 else int(rng.integers(1, 120))
      w = int(rng.integers(1, 700)) if rng.uniform() < 0.3 else int(rng.integers(1, 120))
      big += h * w > 8192
      density = float(rng.choice([0.0, 0.002, 0.05, 0.5, 1.0]))
      mask = rng.random((b, c if rng.integers(2) else 1, h, w)) < density
      height = rng.uniform(-1, 2, (b, 1, h, w)).astype(np.float32)
      pose = np.stack([rng.uniform(-3, 3, b), rng.uniform(-3, 3, b), rng.uniform(-3.2, 3.2, b)], axis=1).astype(np.float32)
      if rng.uniform() < 0.2:
        pose[:, 2] = rng.uniform(-2e-3, 2e-3, b).astype(np.float32)
      proj = dmap.MapProjector(width=64, height=48, hfov=1.2, cam_pose=torch.from_numpy(pose),
                               map_res=float(rng.choice([0.03, 0.05, 0.0625, 0.1])), map_width=w, map_height=h,
                               width_offset=float(rng.uniform(0, w)), height_offset=float(rng.uniform(0, h)),
                               to_global=bool(rng.integers(2)), flip_h=bool(rng.integers(2)), fill_value=fill, reduction=red)
      hm = torch.from_numpy(height).cuda()
      mk = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(mask, (b, c, h, w)))).cuda()
      if semantic:
        top = torch.from_numpy(rng.integers(0, 3, (b, c, h, w)).astype(np.float32)).cuda()
        tms.append(M.TopdownMap(topdown_map=top, mask=mk, height_map=hm.expand(b, c, h, w), map_projector=proj, is_height_map=False))
      else:
        tms.append(M.TopdownMap(topdown_map=hm, mask=mk, height_map=hm, map_projector=proj, is_height_map=True))
    tpose = np.stack([rng.uniform(-3, 3, b), rng.uniform(-3, 3, b), rng.uniform(-3.2, 3.2, b)], axis=1).astype(np.float32)
    target = tms[0].proj.clone(map_res=float(rng.choice([0.04, 0.05, 0.07])), cam_pose=torch.from_numpy(tpose),
                               to_global=bool(rng.integers(2)), flip_h=bool(rng.integers(2)))
    try:
      fast = M.fuse_topdown_maps(*tms, map_projector=target, reduction=red, fill_value=fill)
      want = oracle.fuse_topdown_maps(
          [source_of(m) for m in tms],
          dict(map_res=float(target.map_res), flip_h=bool(target.flip_h), to_global=bool(target.to_global),
               cam_pose=f32(target.cam_pose).reshape(-1, 3), fill_value=fill, reduction=red))
      if want is None:
        empty += 1
        ok = fast.topdown_map is tms[-1].topdown_map or torch.equal(fast.topdown_map, tms[-1].topdown_map)
      else:
        ok = (fast.proj.map_width, fast.proj.map_height) == (want["map_width"], want["map_height"]) and \
            f32(fast.proj.width_offset)[0] == want["woff"] and f32(fast.proj.height_offset)[0] == want["hoff"] and \
            np.array_equal(fast.mask.cpu().numpy(), want["mask"]) and \
            np.array_equal(fast.topdown_map.cpu().numpy(), want["map"]) and \
            np.array_equal(np.broadcast_to(fast.height_map.cpu().numpy(), want["height"].shape), want["height"])
    except Exception as e:      # noqa: BLE001
      ok = False
      print(f"EXCEPTION seed {seed0 + i}: {e!r}", flush=True)
    if not ok:
      bad += 1
      print(f"MISMATCH seed {seed0 + i}: b={b} c={c} semantic={semantic} red={red} maps={[tuple(m.topdown_map.shape) for m in tms]}", flush=True)
  print(f"done: {n} fuse configurations, {bad} with mismatches ({big} source maps of more than 8192 cells, {empty} with no valid cell)", flush=True)

if __name__ == "__main__":
  main()
