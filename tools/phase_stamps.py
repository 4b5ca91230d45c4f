"""Per-phase timing of k_window_scatter from an instrumented build:
    tools/build_variant.sh stamps -DDM_STAMPS
    DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so python tools/phase_stamps.py
Thread 0 of every workgroup records the 100 MHz real-time counter at phase boundaries."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
B, H, W, mh, mw = 64, 480, 640, 512, 512
g = torch.Generator().manual_seed(1234)
depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
pose = torch.empty(B, 3).uniform_(-1, 1, generator=g); pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
lib = _native.lib()
lib.dm_debug_force_legacy_window(1)      # k_window_scatter (host geometry), not the strip path
buf = torch.zeros(4096 * 12, dtype=torch.int64, device="cuda")
lib.dm_debug_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(5):
  top, mask = proj.orth_project(depth, cam_pose=pose)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(-1, 12)
raw = raw[raw[:, 0] != 0]
if raw[:, 7].any():     # exclusive spans: stamps 7..9 sit between 2 and 3
  st = raw[:, [0, 1, 2, 7, 8, 9, 3, 4, 5, 6, 10]]
  names = ["lds init", "first loads", "fill-ahead", "edges + barrier", "spans + barrier", "own rows + hull",
           "scatter loop", "fill rest + barrier", "flush", "drain + ticket"]
  last = raw[raw[:, 11] != 0]
  if len(last):
    d11 = (last[:, 11] - last[:, 10]) * 0.01
    print("last arrivers: %d   border merge us median %.2f max %.2f" % (len(last), np.median(d11), d11.max()))
    print("kernel span incl. border merge us: %.2f" % ((raw[:, [10, 11]].max() - raw[:, 0].min()) * 0.01))
else:
  st = raw[:, :7]
  names = ["lds init", "first loads", "(tables)", "scatter loop", "fill rest + barrier", "flush"]
d = np.diff(st, axis=1).astype(np.float64) * 0.01     # us
print("workgroups: %d   per-WG phase time in us (median / max):" % len(st))
for i, n in enumerate(names):
  print(f"  {n:22s} {np.median(d[:, i]):8.2f} {d[:, i].max():8.2f}")
tot = (st[:, -1] - st[:, 0]) * 0.01
print("  total                  %8.2f %8.2f" % (np.median(tot), tot.max()))
print("kernel span us: %.2f   start skew: %.2f   end skew: %.2f" % (
    (st[:, -1].max() - st[:, 0].min()) * 0.01, (st[:, 0].max() - st[:, 0].min()) * 0.01,
    (st[:, -1].max() - st[:, -1].min()) * 0.01))
# is the end-time skew a per-frame effect (all parts of a frame late together)?
end = (raw[:, 6] - raw[:, 0].min()) * 0.01
if len(end) % 4 == 0:
  pf = end.reshape(-1, 4)
  print("end time us: overall std %.2f   between-frame std %.2f   within-frame std %.2f" % (
      end.std(), pf.mean(axis=1).std(), np.sqrt(((pf - pf.mean(axis=1, keepdims=True)) ** 2).mean())))
  print("by part index (mean end us):", np.round(pf.mean(axis=0), 2))
  xcd = np.arange(len(end)) % 8
  print("by XCD (mean end us):", np.round([end[xcd == i].mean() for i in range(8)], 2))
