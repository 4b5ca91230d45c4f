"""Per-phase timing of k_strip_scatter from an instrumented build:
    tools/build_variant.sh stamps -DDM_STAMPS
    DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so python tools/strip_stamps.py
Thread 0 of every workgroup records the 100 MHz real-time counter at phase boundaries."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
B, H, W, mh, mw = [int(v) for v in os.environ.get("DM_STAMPS_SHAPE", "64,480,640,512,512").split(",")]
g = torch.Generator().manual_seed(1234)
ROT = int(os.environ.get("DM_STAMPS_ROT", "1"))     # depth batches / output blocks in rotation (> 256 MiB live: HBM-served)
depths = [torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda() for _ in range(ROT)]
depth = depths[0]
if os.environ.get("DM_STAMPS_FILL_SPLIT"):
  _native.lib().dm_debug_fill_split(int(os.environ["DM_STAMPS_FILL_SPLIT"]))
if os.environ.get("DM_STAMPS_NT"):
  _native.lib().dm_debug_force_nt_fill(int(os.environ["DM_STAMPS_NT"]))
pose = torch.empty(B, 3).uniform_(-1, 1, generator=g); pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
lib = ctypes.CDLL(_native.LIB_PATH)
_native.lib()
buf = torch.zeros(4096 * 12 + 4096 * 16, dtype=torch.int64, device="cuda")
lib.dm_debug_strip_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
if os.environ.get("DM_STAMPS_ONE"):
  lib.dm_debug_one_kernel(int(os.environ["DM_STAMPS_ONE"]))
  _native.lib().dm_debug_one_kernel(int(os.environ["DM_STAMPS_ONE"]))
if os.environ.get("DM_STAMPS_LEGACY"):
  _native.lib().dm_debug_force_legacy_window(1)
if os.environ.get("DM_STAMPS_STRIPS"):
  _native.lib().dm_debug_force_strips(int(os.environ["DM_STAMPS_STRIPS"]))
C = int(os.environ.get("DM_STAMPS_CHANNELS", "0"))
value = None
if C:      # value maps: the stamps then are those of the value pass (the kernel that runs last)
  value = torch.nn.functional.one_hot(torch.randint(0, C, (B, H, W), generator=g), C).permute(0, 3, 1, 2).float().contiguous().cuda()
  proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                           width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                           trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=0.0)
  buf = torch.zeros(4096 * 12 * C + 4096 * 16 * C, dtype=torch.int64, device="cuda")
  lib.dm_debug_strip_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
keep, count = [None] * (ROT + 1), [0]
def run():
  j = count[0]; count[0] += 1
  keep[j % len(keep)] = None
  keep[j % len(keep)] = proj.orth_project(depths[j % ROT], value_map=value, cam_pose=pose)
  return keep[j % len(keep)]
for _ in range(5):
  top, mask = run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
  top, mask = run()
ev[1].record(); torch.cuda.synchronize()
print("orth_project call (pose upload + prepare + scatter + combine): %.1f us" % (ev[0].elapsed_time(ev[1]) * 50))
allb = buf.cpu().numpy()
nwg = 256 * C if C else 4096       # (value maps: C <= 16, so that the workgroup stamps stay below the wave stamps)
raw = allb[:nwg * 12].reshape(-1, 12)
waves = allb[4096 * 12:4096 * 12 + 4096 * 16].reshape(-1, 16)[:(raw[:, 0] != 0).sum()] if not C else np.zeros((0, 16))
raw = raw[raw[:, 0] != 0]
st = raw[:, :7]
def seg(name, i, j):
  dd = (raw[:, j] - raw[:, i]) * 0.01
  print(f"  {name:44s} {np.median(dd):8.2f} {dd.max():8.2f}")
print("workgroups: %d   per-WG phase time in us (median / max):" % len(st))
seg("(thread 0 is in wave 0: start -> in front of the barrier)", 0, 1)
TAIL = bool(os.environ.get("DM_STAMPS_TAIL"))      # the one-kernel form: stamps 7 .. 10 are the tail's
if TAIL:
  seg("head (start -> second loads requested)", 0, 11)
elif raw[:, 7].any() and raw[:, 8].any():
  seg("wave 0: kernel arguments + pose record loaded", 0, 9)
  seg("wave 0: geometry", 9, 10)
  seg("wave 0: -> barrier passed", 10, 7)
  seg("row tables", 7, 8)
  seg("loop scalars reloaded + barrier", 8, 2)
else:
  seg("row entries -> lds, barrier", 1, 2)
if not TAIL:
  seg("second loads (head of the pipeline)", 2, 11)
seg("pixel loop", 11, 4)
seg("rest of the fill duty + barrier", 4, 5)
if TAIL:
  seg("tail: shared groups -> slab", 5, 7)
  seg("tail: drain + barrier", 7, 8)
  seg("tail: owned groups -> map (counter in flight)", 8, 9)
  seg("tail: counter back + barrier", 9, 10)
  seg("tail: combine (the last workgroup of a frame)", 10, 6)
else:
  seg("flush", 5, 6)
tot = (st[:, -1] - st[:, 0]) * 0.01
print("  total                              %8.2f %8.2f" % (np.median(tot), tot.max()))
print("kernel span us: %.2f   start skew: %.2f   end skew: %.2f" % (
    (st[:, -1].max() - st[:, 0].min()) * 0.01, (st[:, 0].max() - st[:, 0].min()) * 0.01,
    (st[:, -1].max() - st[:, -1].min()) * 0.01))

if waves.size and waves.any():
  rel = (waves - st[:, :1]) * 0.01            # each wave's loop end, relative to its workgroup's start
  print("loop end by wave index (mean us after WG start):", np.round(rel.mean(axis=0), 1))
  print("  first / last wave of a WG (median): %.2f / %.2f" % (np.median(rel.min(axis=1)), np.median(rel.max(axis=1))))

# where the slow workgroups are: pixel-loop time by strip, by XCD (workgroup id mod 8), by frame
loop = (raw[:, 4] - raw[:, 11]) * 0.01
wg = np.arange(len(loop))
if len(loop) % 4 == 0 and not os.environ.get("DM_STAMPS_STRIPS") and not C:
  print("pixel loop by strip:", np.round([loop[wg % 4 == s].mean() for s in range(4)], 2))
  print("pixel loop by XCD:  ", np.round([loop[wg % 8 == x].mean() for x in range(8)], 2))
  fr = loop.reshape(-1, 4).mean(axis=1)
  yaw = pose[:, 2].numpy()
  order = np.argsort(fr)
  print("slowest frames (loop us, yaw deg):", [(round(float(fr[i]), 1), int(np.degrees(yaw[i]))) for i in order[-6:]])
  print("fastest frames (loop us, yaw deg):", [(round(float(fr[i]), 1), int(np.degrees(yaw[i]))) for i in order[:6]])
  print("per-WG loop: p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(loop, [10, 50, 90, 100])))
  start = (raw[:, 0] - raw[:, 0].min()) * 0.01
  print("corr(loop, start time) = %.2f" % np.corrcoef(loop, start)[0, 1])

if len(loop) % 4 == 0 and len(loop) == 4 * B and not os.environ.get("DM_STAMPS_STRIPS") and not C:
  # does a workgroup's loop time follow its LDS window's row stride?  (row stride mod 32 words: the banks a
  # column of cells falls into)
  from dungeon_maps_amd import functional as Fn
  cam = proj.cam_params
  call = Fn._Call(depth, None, None, pose, mw / 2., mh / 2., proj.cam_pitch, proj.cam_height, 0.03, mw, mh, cam.fx, cam.fy,
                  cam.cx, cam.cy, 0.15, 5.05, None, None, True, True, -np.inf, None, None)
  geom = np.zeros((B, 8 + 4 * 8), dtype=np.int32)
  P = _native.lib().dm_debug_strip_geometry(ctypes.byref(call.params), ctypes.c_void_p(call.frames.data_ptr()),
                                            geom.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), None, None)
  if P == 4:
    ww = geom[:, 8:8 + 16].reshape(B, 4, 4)[:, :, 2].reshape(-1)        # window width of workgroup (frame, strip)
    hh = geom[:, 8:8 + 16].reshape(B, 4, 4)[:, :, 3].reshape(-1)
    for m in (32, 16, 8):
      print("pixel loop by window width mod %d:" % m, {int(k): (round(float(loop[ww % m == k].mean()), 2), int((ww % m == k).sum()))
                                                       for k in sorted(set((ww % m).tolist()))})
    print("corr(loop, window cells) = %.2f   corr(loop, window width) = %.2f   corr(loop, window rows) = %.2f" % (
        np.corrcoef(loop, ww * hh)[0, 1], np.corrcoef(loop, ww)[0, 1], np.corrcoef(loop, hh)[0, 1]))

if os.environ.get("DM_STAMPS_SLOWEST"):
  # the slowest workgroups of the launch: (index, frame, strip, index % 8) and their phases
  order = np.argsort(-tot)[:int(os.environ["DM_STAMPS_SLOWEST"])]
  t0 = st[:, 0].min()
  print("slowest workgroups: index frame strip idx%8 | start head loop rest flush total (us)")
  for i in order:
    print("  %4d %3d %d %d | %5.2f %5.2f %5.2f %5.2f %5.2f %6.2f" % (
        i, i // 4, i % 4, i % 8, (st[i, 0] - t0) * 0.01, (raw[i, 11] - raw[i, 0]) * 0.01, (raw[i, 4] - raw[i, 11]) * 0.01,
        (raw[i, 5] - raw[i, 4]) * 0.01, (raw[i, 6] - raw[i, 5]) * 0.01, tot[i]))
