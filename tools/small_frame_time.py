"""BASELINE configs[0] (B = 1, 320x240 -> 256x256, the reference's demo): time per orth_project call, per
MapBuilder.step(merge=True), and which kernels they launch (run under rocprofv3 --kernel-trace for those)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
lib = _native.lib()
B, H, W, mh, mw = 1, 240, 320, 256, 256
g = torch.Generator().manual_seed(1)
d = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
poses = [torch.tensor([[0.1 * i, 0.05 * i, 0.1 * i]]) for i in range(8)]
def timed(fn, n=200):
  for i in range(10): fn(i)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter(); e0.record()
  for i in range(n): fn(i)
  e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
  return e0.elapsed_time(e1) * 1e3 / n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for strips in (0, 1, 2, 4, 8):
  lib.dm_debug_force_strips(strips)
  ev, host, wall = timed(lambda i: proj.orth_project(d, cam_pose=poses[i % 8]))
  split = (4 * __import__("ctypes").c_int32)(); lib.dm_debug_last_split(split)
  print("orth_project, forced strips %d -> path %d split %s: %.1f us/call (events), host %.1f, wall %.1f" % (
      strips, lib.dm_debug_last_path(), list(split)[:3], ev, host, wall))
lib.dm_debug_force_strips(0)
lib.dm_debug_force_legacy_window(1)
ev, host, wall = timed(lambda i: proj.orth_project(d, cam_pose=poses[i % 8]))
print("orth_project, window path: %.1f us/call (events), host %.1f, wall %.1f" % (ev, host, wall))
lib.dm_debug_force_legacy_window(0)
builder = dmap.MapBuilder(proj)
def step(i):
  builder.step(depth_map=d, cam_pose=poses[i % 8], merge=True)
ev, host, wall = timed(step, 100)
print("MapBuilder.step(merge=True): %.1f us/frame (events), host %.1f, wall %.1f" % (ev, host, wall))
