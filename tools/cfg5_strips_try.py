import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
lib = _native.lib()
B, H, W, mh, mw = 16, 960, 1280, 2048, 2048
ROT = 3
g = torch.Generator().manual_seed(1234)
depths = [torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda() for _ in range(ROT)]
pose = torch.empty(B, 3).uniform_(-1, 1, generator=g); pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
keep = [None] * (ROT + 1)
def call(j):
  keep[j % len(keep)] = None
  keep[j % len(keep)] = proj.orth_project(depths[j % ROT], cam_pose=pose)
def b2b(n=32):
  for j in range(8): call(j)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for j in range(n): call(8 + j)
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / n
ref = None
for strips in (0, 8, 4, 0):
  lib.dm_debug_force_strips(strips)
  t = min(b2b() for _ in range(3))
  top, mask = proj.orth_project(depths[0], cam_pose=pose)
  split = (ctypes.c_int32 * 4)(); lib.dm_debug_last_split(split)
  if ref is None: ref = (top.clone(), mask.clone())
  print("forced strips %d: path %d split %s: %.1f us per call, same=%s" % (strips, lib.dm_debug_last_path(), list(split)[:3], t,
        torch.equal(top, ref[0]) and torch.equal(mask, ref[1])))
lib.dm_debug_force_strips(0)
