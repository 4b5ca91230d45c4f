"""Times dm_orth_project_f32 on the LDS-windowed path and on the generic (global atomic) path for a
list of shapes and prints the split the windowed path chose (depth bands)."""
import sys, os, time, ctypes, numpy as np, torch
sys.path.insert(0, os.getcwd())
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
lib = _native.lib()
def run(B, H, W, mh, mw, res, C=0):
  g = torch.Generator().manual_seed(5)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
  pose = torch.empty(B, 3).uniform_(-1, 1, generator=g); pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
  proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                           width_offset=mw / 2., height_offset=mh / 2., map_res=res, map_width=mw, map_height=mh,
                           trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
  val = torch.randn(B, C, H, W, device="cuda") if C else None
  ts = []
  for force in (0, 1):
    lib.dm_debug_force_generic_path(force)
    for _ in range(5): proj.orth_project(depth, value_map=val, cam_pose=pose)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): proj.orth_project(depth, value_map=val, cam_pose=pose)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e6)
    if force == 0:
      out = (ctypes.c_int32 * 4)(); lib.dm_debug_last_split(out)
  lib.dm_debug_force_generic_path(0)
  print((B, H, W, mh, mw, res, C), list(out)[:3], "window %.1f us  generic %.1f us" % tuple(ts), flush=True)
run(16, 960, 1280, 2048, 2048, 0.01)
run(8, 240, 320, 2048, 2048, 0.01)
run(12, 240, 320, 2048, 2048, 0.008)
run(1, 480, 640, 1024, 1024, 0.01)
run(4, 480, 640, 1024, 1024, 0.01)
run(1, 960, 1280, 2048, 2048, 0.01)
run(4, 960, 1280, 2048, 2048, 0.01)
run(64, 480, 640, 2048, 2048, 0.01)
run(16, 960, 1280, 2048, 2048, 0.02)
run(4, 960, 1280, 1024, 1024, 0.02)
run(1, 480, 640, 512, 512, 0.03)
run(1, 960, 1280, 1024, 1024, 0.03)
