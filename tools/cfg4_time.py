"""cfg4 per rank (64 frames of one trajectory -> one 1024x1024 map) through dm_orth_project_fused_f32, eight depth
batches in rotation (HBM-served): time per call by HIP events, calls back to back; a checksum of the result."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
B, H, W, mh, mw = 64, 480, 640, 1024, 1024
ROT = int(os.environ.get("DM_ROT", "8"))
g = torch.Generator().manual_seed(1234)
depths = [torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda() for _ in range(ROT)]
k = torch.arange(B, dtype=torch.float32)
pose = torch.stack((0.02 * k, 0.01 * k, 0.01 * k), dim=1)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
for j in range(16):
  out = proj.orth_project_fused(depths[j % ROT], cam_pose=pose)
torch.cuda.synchronize()
import time
res, host = [], []
for rep in range(3):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  t0 = time.perf_counter()
  for j in range(64):
    out = proj.orth_project_fused(depths[j % ROT], cam_pose=pose)
  host.append((time.perf_counter() - t0) * 1e6 / 64)
  e1.record(); torch.cuda.synchronize()
  res.append(e0.elapsed_time(e1) * 1e3 / 64)
top = proj.orth_project_fused(depths[0], cam_pose=pose)
fin = torch.where(torch.isfinite(top[0]), top[0], torch.zeros_like(top[0]))
print("%s: %s us/call (host %s)  checksum %.6f mask %d" % (os.environ.get("DUNGEON_MAPS_AMD_LIB", "default").split("/")[-1],
      ["%.1f" % r for r in res], ["%.1f" % r for r in host], float(fin.double().sum()), int(top[1].sum())))
