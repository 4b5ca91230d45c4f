"""What a pair of HIP events around ONE launch sequence costs, by event flags (cfg2, prepared frames):
the bracketed launch time against the back-to-back figure bench.py reports."""
import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
hip = ctypes.CDLL("libamdhip64.so")
lib = _native.lib()
B, H, W, mh, mw = 64, 480, 640, 512, 512
g = torch.Generator().manual_seed(1234)
depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
pose = torch.empty(B, 3).uniform_(-1, 1, generator=g); pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
prep = proj.prepare(B, cam_pose=pose)
outs = (torch.empty((B, 1, mh, mw), device="cuda"), torch.empty((B, 1, mh, mw), dtype=torch.bool, device="cuda"))
for _ in range(5): prep.orth_project(depth, out=outs)
torch.cuda.synchronize()
for flags, name in ((0, "default"), (0x20000000, "DisableSystemFence"), (0x40000000, "ReleaseToDevice")):
  n = 100
  ea = [ctypes.c_void_p() for _ in range(n)]; eb = [ctypes.c_void_p() for _ in range(n)]
  for e in ea + eb:
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
  for i in range(n):
    lib.dm_debug_record_before_projection(ea[i]); lib.dm_debug_record_after_projection(eb[i])
    prep.orth_project(depth, out=outs)
  torch.cuda.synchronize()
  ms = ctypes.c_float(); tot = []
  for i in range(n):
    assert hip.hipEventElapsedTime(ctypes.byref(ms), ea[i], eb[i]) == 0
    tot.append(ms.value * 1e3)
  print(name, "bracketed launch us: mean %.2f median %.2f" % (np.mean(tot), np.median(tot)))
