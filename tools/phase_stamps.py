"""Read the per-phase s_memtime stamps an instrumented build (libdm_stamps.so) leaves in `out`
(the merge kernel is skipped by reading stamps from a run whose merge overwrites only U)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
B, H, W, mh, mw = 64, 480, 640, 512, 512
g = torch.Generator().manual_seed(1234)
depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
pose = torch.empty(B, 3).uniform_(-1, 1, generator=g); pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
for _ in range(5):
  top, mask = proj.orth_project(depth, cam_pose=pose)
torch.cuda.synchronize()
st = top.view(torch.int64).flatten()[:256 * 8].cpu().numpy().reshape(256, 8)
d = np.diff(st[:, :6], axis=1).astype(np.float64)   # s_memtime ticks (100 MHz? or shader clock)
names = ["init", "scatter+fill", "fill tail", "barrier", "flush"]
print("per-WG phase ticks (median / max):")
for i, n in enumerate(names):
  print(f"  {n:14s} {np.median(d[:, i]):10.0f} {d[:, i].max():10.0f}")
tot = (st[:, 5] - st[:, 0]).astype(np.float64)
print("  total          %10.0f %10.0f" % (np.median(tot), tot.max()))
print("kernel span ticks:", st[:, 5].max() - st[:, 0].min(), " start skew:", st[:, 0].max() - st[:, 0].min())
