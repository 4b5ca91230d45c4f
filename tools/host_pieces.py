"""Host time of the pieces of one MapProjector.orth_project(depth, cam_pose=...) call (no profiler: each
piece timed alone over many repetitions; the GPU is drained between pieces)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native, frames, functional as F
lib = _native.lib()
def t(fn, n=2000):
  for _ in range(50): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n): fn()
  dt = time.perf_counter() - t0
  torch.cuda.synchronize()
  return dt / n * 1e6
for B, H, W, mh, mw in ((1, 240, 320, 256, 256), (64, 480, 640, 512, 512)):
  g = torch.Generator().manual_seed(1)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
  pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
  proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                           width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                           trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
  n = 300 if B == 64 else 2000
  print("B = %d" % B)
  print("  whole call                 %6.1f us" % t(lambda: proj.orth_project(depth, cam_pose=pose), n))
  print("  build_frame_table          %6.1f us" % t(lambda: frames.build_frame_table(B, pose, proj.cam_pitch, proj.cam_height, mw / 2., mh / 2.)))
  cam = proj.cam_params
  mk = lambda: F._Call(depth, None, None, pose, mw / 2., mh / 2., proj.cam_pitch, proj.cam_height, 0.03, mw, mh,
                       cam.fx, cam.fy, cam.cx, cam.cy, 0.15, 5.05, None, None, True, True, -np.inf, None, None)
  print("  _Call (incl. frame table)  %6.1f us" % t(mk))
  call = mk()
  p = call.params
  shape = (B, 1, mh, mw)
  print("  3 x torch.empty            %6.1f us" % t(lambda: (torch.empty(shape, dtype=torch.float32, device="cuda"), torch.empty(shape, dtype=torch.bool, device="cuda"), torch.empty(call.ws_bytes, dtype=torch.uint8, device="cuda"))))
  top = torch.empty(shape, dtype=torch.float32, device="cuda"); mask = torch.empty(shape, dtype=torch.bool, device="cuda")
  ws = torch.empty(call.ws_bytes, dtype=torch.uint8, device="cuda")
  stream = F._stream_ptr(depth.device)
  def native():
    lib.dm_orth_project_f32(ctypes.byref(p), call.frames.data_ptr(), depth.data_ptr(), None, None, top.data_ptr(),
                            mask.data_ptr(), None, None, None, ws.data_ptr(), call.ws_bytes, _native.status_ptr(), stream)
  print("  native call (ctypes + C)   %6.1f us" % t(native, n))
  print("  check_status               %6.1f us" % t(_native.check_status))
  print("  _stream_ptr                %6.1f us" % t(lambda: F._stream_ptr(depth.device)))
