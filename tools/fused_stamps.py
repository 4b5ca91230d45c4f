"""Per-phase timing of k_strip_fused (dm_orth_project_fused_f32, BASELINE configs[3]) from the -DDM_STAMPS build:
    tools/build_variant.sh stamps -DDM_STAMPS
    DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so python tools/fused_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
B, H, W, mh, mw = 64, 480, 640, 1024, 1024
ROT = int(os.environ.get("DM_STAMPS_ROT", "8"))
g = torch.Generator().manual_seed(1234)
depths = [torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda() for _ in range(ROT)]
k = torch.arange(B, dtype=torch.float32)
pose = torch.stack((0.02 * k, 0.01 * k, 0.01 * k), dim=1)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
lib = ctypes.CDLL(_native.LIB_PATH)
_native.lib()
buf = torch.zeros(4096 * 12 + 4096 * 16, dtype=torch.int64, device="cuda")
lib.dm_debug_strip_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
for j in range(6):
  proj.orth_project_fused(depths[j % ROT], cam_pose=pose)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for j in range(32):
  proj.orth_project_fused(depths[j % ROT], cam_pose=pose)
e1.record(); torch.cuda.synchronize()
print("orth_project_fused call: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 32))
split = (ctypes.c_int32 * 4)(); _native.lib().dm_debug_last_fused_split(split); print("split (wp, P, F, G):", list(split))
raw = buf.cpu().numpy()[:4096 * 12].reshape(-1, 12)
raw = raw[raw[:, 0] != 0]
names = ["start -> geometry / LDS init done", "barrier", "window, first loop scalars", "pixel loop", "spans", "barrier", "flush"]
print("workgroups: %d   per-WG phase time in us (median / max):" % len(raw))
for i, n in enumerate(names):
  dd = (raw[:, i + 1] - raw[:, i]) * 0.01
  print(f"  {n:40s} {np.median(dd):8.2f} {dd.max():8.2f}")
tot = (raw[:, 7] - raw[:, 0]) * 0.01
print("  total                                    %8.2f %8.2f" % (np.median(tot), tot.max()))
print("kernel span us: %.2f   start skew: %.2f   end skew: %.2f" % ((raw[:, 7].max() - raw[:, 0].min()) * 0.01,
      (raw[:, 0].max() - raw[:, 0].min()) * 0.01, (raw[:, 7].max() - raw[:, 7].min()) * 0.01))
order = np.argsort(-tot)[:8]
print("slowest workgroups: index | " + " ".join(n.split()[0] for n in names) + " total")
for i in order:
  print("  %4d | %s %6.2f" % (i, " ".join("%5.2f" % ((raw[i, j + 1] - raw[i, j]) * 0.01) for j in range(7)), tot[i]))
