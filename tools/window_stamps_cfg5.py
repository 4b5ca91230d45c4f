"""Per-phase timing of k_window_scatter at BASELINE configs[4]'s shape (16 x 1280x960 -> 2048x2048) from the -DDM_STAMPS
build, three depth batches in rotation (HBM-served):
    DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so python tools/window_stamps_cfg5.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
B, H, W, mh, mw = [int(v) for v in os.environ.get("DM_STAMPS_SHAPE", "16,960,1280,2048,2048").split(",")]
ROT = int(os.environ.get("DM_STAMPS_ROT", "3"))
g = torch.Generator().manual_seed(1234)
depths = [torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda() for _ in range(ROT)]
pose = torch.empty(B, 3).uniform_(-1, 1, generator=g); pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
lib = _native.lib()
buf = torch.zeros(4096 * 12, dtype=torch.int64, device="cuda")
lib_raw = ctypes.CDLL(_native.LIB_PATH)
lib_raw.dm_debug_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
keep = [None] * (ROT + 1)
for j in range(8):
  keep[j % len(keep)] = None
  keep[j % len(keep)] = proj.orth_project(depths[j % ROT], cam_pose=pose)
torch.cuda.synchronize()
split = (ctypes.c_int32 * 4)(); lib.dm_debug_last_split(split); print("path", lib.dm_debug_last_path(), "split (pc, pr, pd):", list(split)[:3])
raw = buf.cpu().numpy().reshape(-1, 12)
raw = raw[raw[:, 0] != 0]
st = raw[:, :7]
names = ["lds init", "first loads", "(tables)", "scatter loop", "fill rest + barrier", "flush"]
d = np.diff(st, axis=1).astype(np.float64) * 0.01
print("workgroups: %d   per-WG phase time in us (median / max):" % len(st))
for i, n in enumerate(names):
  print(f"  {n:22s} {np.median(d[:, i]):8.2f} {d[:, i].max():8.2f}")
tot = (st[:, -1] - st[:, 0]) * 0.01
print("  total                  %8.2f %8.2f" % (np.median(tot), tot.max()))
print("kernel span us: %.2f   start skew: %.2f   end skew: %.2f" % ((st[:, -1].max() - st[:, 0].min()) * 0.01,
      (st[:, 0].max() - st[:, 0].min()) * 0.01, (st[:, -1].max() - st[:, -1].min()) * 0.01))
order = np.argsort(-tot)[:10]
t0 = st[:, 0].min()
print("slowest workgroups: index | start " + " ".join(n.split()[0] for n in names) + " total")
for i in order:
  print("  %4d | %6.2f %s %7.2f" % (i, (st[i, 0] - t0) * 0.01, " ".join("%6.2f" % d[i, j] for j in range(6)), tot[i]))
order = np.argsort(tot)[:4]
print("fastest:")
for i in order:
  print("  %4d | %6.2f %s %7.2f" % (i, (st[i, 0] - t0) * 0.01, " ".join("%6.2f" % d[i, j] for j in range(6)), tot[i]))
