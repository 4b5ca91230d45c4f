"""cfg4 per rank (64 frames of one trajectory -> one 1024x1024 map) through dm_orth_project_fused_f32 with the
strip width and the frames per workgroup forced (dm_debug_force_fused_split): time per call by HIP events."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
lib = _native.lib()
B, H, W, mh, mw = 64, 480, 640, 1024, 1024
g = torch.Generator().manual_seed(1)
ROT = int(os.environ.get("DM_ROT", "8"))       # depth batches in rotation: HBM-served
depths = [torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda() for _ in range(ROT)]
depth = depths[0]
k = torch.arange(B, dtype=torch.float32)
pose = torch.stack((0.02 * k, 0.01 * k, 0.01 * k), dim=1)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
ref = None
for wp, F in [(0, 0), (160, 1), (80, 1), (80, 2), (40, 2), (40, 4), (32, 4), (20, 4), (20, 8), (24, 8), (40, 8), (16, 8)]:
  lib.dm_debug_force_fused_split(wp, F)
  try:
    for _ in range(3):
      out = proj.orth_project_fused(depth, cam_pose=pose)
    split = (ctypes.c_int32 * 4)(); lib.dm_debug_last_fused_split(split)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(32):
      out = proj.orth_project_fused(depths[j % ROT], cam_pose=pose)
    e1.record(); torch.cuda.synchronize()
    out = proj.orth_project_fused(depth, cam_pose=pose)
    if ref is None: ref = out
    same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    print("wp %3d F %d -> split %s  %.1f us/call  same=%s" % (wp, F, list(split), e0.elapsed_time(e1) * 1e3 / 32, same))
  finally:
    lib.dm_debug_force_fused_split(0, 0)
