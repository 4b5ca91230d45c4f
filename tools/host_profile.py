"""Where does the host time of one orth_project call go?  (run on the GPU box)"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native, frames, functional as F

B, H, W, mh, mw = 64, 480, 640, 512, 512
g = torch.Generator().manual_seed(0)
depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
                         cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2.,
                         map_res=0.03, map_width=mw, map_height=mh, trunc_depth_min=0.15,
                         trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)

def t(fn, n=300):
  for _ in range(20): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n): fn()
  dt = (time.perf_counter() - t0) / n * 1e6
  torch.cuda.synchronize()
  return dt

print("full API call            %7.1f us" % t(lambda: proj.orth_project(depth, cam_pose=pose)))
print("orth_project_and_fuse    %7.1f us" % t(lambda: proj.orth_project_and_fuse(depth, cam_pose=pose)))
print("functional (no forward)  %7.1f us" % t(lambda: F.orth_project(depth, None, None, pose, 256., 256., proj.cam_pitch, proj.cam_height, 0.03, mw, mh,
               proj.cam_params.fx, proj.cam_params.fy, proj.cam_params.cx, proj.cam_params.cy,
               0.15, 5.05, None, None, True, True, -np.inf, None)))
print("build_frame_table        %7.1f us" % t(lambda: frames.build_frame_table(B, pose, proj.cam_pitch, proj.cam_height, 256., 256.)))
call = F._Call(depth, None, None, pose, 256., 256., proj.cam_pitch, proj.cam_height, 0.03, mw, mh,
               proj.cam_params.fx, proj.cam_params.fy, proj.cam_params.cx, proj.cam_params.cy,
               0.15, 5.05, None, None, True, True, -np.inf, None, None)
print("_Call construction       %7.1f us" % t(lambda: F._Call(depth, None, None, pose, 256., 256., proj.cam_pitch, proj.cam_height, 0.03, mw, mh,
               proj.cam_params.fx, proj.cam_params.fy, proj.cam_params.cx, proj.cam_params.cy,
               0.15, 5.05, None, None, True, True, -np.inf, None, None)))
p = call.params
top = torch.empty((B, 1, mh, mw), device="cuda"); mask = torch.empty((B, 1, mh, mw), dtype=torch.bool, device="cuda")
ws, wsb = call.workspace()
lib = _native.lib()
stream = torch.cuda.current_stream().cuda_stream
def native():
  lib.dm_orth_project_f32(ctypes.byref(p), call.frames.data_ptr(), depth.data_ptr(), None, None,
                          top.data_ptr(), mask.data_ptr(), None, None, None, ws.data_ptr(), wsb, None, stream)
print("native call only         %7.1f us  (window calc + memcpy + 2 launches)" % t(native))
print("3x torch.empty           %7.1f us" % t(lambda: (torch.empty((B, 1, mh, mw), device="cuda"), torch.empty((B, 1, mh, mw), dtype=torch.bool, device="cuda"), torch.empty(wsb, dtype=torch.uint8, device="cuda"))))
print("workspace_bytes call     %7.1f us" % t(lambda: lib.dm_orth_project_workspace_bytes(ctypes.byref(p))))
print("fuse_batch               %7.1f us" % t(lambda: dmap.fuse_batch(top)))
print("mask_from_map            %7.1f us" % t(lambda: dmap.mask_from_map(top[0], -np.inf)))
torch.cuda.synchronize()
# GPU time of the native sequence when the queue is kept full
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(10): native()
torch.cuda.synchronize(); ev0.record()
for _ in range(200): native()
ev1.record(); torch.cuda.synchronize()
print("native sequence, GPU-side %6.1f us per call (200 back-to-back)" % (ev0.elapsed_time(ev1) * 1e3 / 200))
