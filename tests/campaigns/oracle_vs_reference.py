"""One-off (build container only: imports /root/reference): the oracle against the
reference itself on many seeded random configurations -- a wider pin than the committed
fixtures.  Usage: python tests/campaigns/oracle_vs_reference.py [first] [count]"""
import importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tests/golden/gen_golden.py"))
gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
from oracle import oracle
from conftest import project_kwargs

torch.set_num_threads(1)
ref = gg._import_reference()

def one(seed):
  rng = np.random.default_rng(70_000 + seed)
  B = int(rng.choice([1, 2, 3]))
  h, w = [(24, 32), (40, 56), (48, 64), (30, 50)][int(rng.integers(4))]
  mh, mw = [(48, 64), (64, 48), (56, 56), (80, 80)][int(rng.integers(4))]
  hfov = float(rng.uniform(0.6, 2.0))
  pitch = rng.uniform(-0.9, 0.5, size=B).astype(np.float32)
  camh = rng.uniform(0.2, 2.0, size=B).astype(np.float32)
  if rng.integers(2):
    depth = np.stack([gg.scene_depth(rng, h, w, hfov, float(pitch[b]), float(camh[b])) for b in range(B)])[:, None]
  else:
    depth = rng.uniform(0.1, 8.0, size=(B, 1, h, w)).astype(np.float32)
  if rng.integers(8) == 0:      # a few special values
    depth.reshape(-1)[rng.integers(0, depth.size, 6)] = [np.nan, np.inf, -np.inf, 0.0, -1.0, 1e30]
  pose = np.stack([rng.uniform(-2, 2, B), rng.uniform(-2, 2, B), rng.uniform(-np.pi, np.pi, B)], axis=1).astype(np.float32)
  is_max = bool(rng.integers(4))
  C = int(rng.choice([0, 0, 3]))
  value = rng.normal(size=(B, C, h, w)).astype(np.float32) if C else None
  cfg = dict(width=w, height=h, hfov=hfov, vfov=None if rng.integers(2) else float(rng.uniform(0.5, 1.6)),
             map_res=float(rng.choice([0.03, 0.05, 0.08, 0.1, 1.0 / 3])), map_width=mw, map_height=mh,
             trunc_depth_min=None if rng.integers(6) == 0 else float(rng.choice([0.0, 0.15, 0.5])),
             trunc_depth_max=None if rng.integers(6) == 0 else float(rng.choice([1.5, 2.5, 5.05, 7.0])),
             trunc_height_max=None if rng.integers(3) else float(rng.uniform(0.2, 1.2)),
             clip_border=int(rng.choice([0, 0, 3])), to_global=bool(rng.integers(2)),
             flip_h=bool(rng.integers(4)),
             fill_value=(-np.inf if is_max else np.inf) if rng.integers(3) else float(rng.uniform(-1, 1)),
             reduction="max" if is_max else "min")
  woff = (mw / 2 + rng.uniform(-20, 20, size=B)).astype(np.float32)
  hoff = (mh / 2 + rng.uniform(-20, 20, size=B)).astype(np.float32)
  valid = (rng.uniform(size=(B, 1, h, w)) > 0.1) if rng.integers(3) == 0 else None
  tops, masks = [], []
  for b in range(B):
    call = dict(cam_pose=pose[b], cam_pitch=pitch[b:b + 1], cam_height=camh[b:b + 1],
                width_offset=woff[b:b + 1], height_offset=hoff[b:b + 1])
    r = gg.run_orth(ref, cfg, depth[b], value=None if value is None else value[b],
                    valid=None if valid is None else valid[b], call_kwargs=call, intermediates=False)
    tops.append(r["topdown"]); masks.append(r["mask"])
  want_t, want_m = np.concatenate(tops, axis=0), np.concatenate(masks, axis=0)
  kw = project_kwargs(cfg, oracle.camera_intrinsics)
  kw.update(cam_pose=pose, cam_pitch=pitch, cam_height=camh, width_offset=woff, height_offset=hoff)
  top, mask = oracle.orth_project(depth, value_map=value, valid_map=valid, **kw)
  bm = int((mask != want_m).sum())
  bv = int((~((top == want_t) | (np.isnan(top) & np.isnan(want_t)))).sum())
  return bm, bv

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for s in range(first, first + count):
  bm, bv = one(s)
  if bm or bv:
    bad += 1
    print("MISMATCH seed", s, "mask cells", bm, "map cells", bv, flush=True)
print("done: %d configurations, %d with mismatches" % (count, bad))
