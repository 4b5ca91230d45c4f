"""The reference demo's loop (demos/height_map/run.py:93-145) on synthetic frames: build.step(merge=False) with the demo's
keywords + build.merge(local_map) per frame; and cfg1 (B = 1, 320x240 -> 256x256): orth_project / plot / merge / step."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
def t(fn, n=400):
  for i in range(30): fn(i)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(n): fn(i)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e6
g = torch.Generator().manual_seed(1)
# --- the demo
W, H = 800, 600
proj = dmap.MapProjector(width=W, height=H, hfov=math.radians(70), vfov=None, cam_pose=[0., 0., 0.], width_offset=0.,
                         height_offset=0., cam_pitch=-0.3490659, cam_height=0.88, map_res=0.03, map_width=600,
                         map_height=600, trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=50,
                         fill_value=-np.inf, to_global=True)
build = dmap.MapBuilder(map_projector=proj)
depth = torch.empty(1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
poses = [np.array([0.05 * math.sin(0.3 * i), 0.05 * i % 1.0, 0.2 * i], dtype=np.float32) for i in range(16)]
def demo_step(i):
  local_map = build.step(depth_map=depth, cam_pose=poses[i % 16], to_global=False, map_res=0.015,
                         width_offset=build.proj.map_width / 2., height_offset=0., map_width=600, map_height=600,
                         center_mode=dmap.CenterMode.none, merge=False)
  return local_map
def demo_frame(i):
  build.merge(demo_step(i), keep_pose=False)
print("demo: step(merge=False)          %7.1f us" % t(demo_step))
build.reset()
print("demo: step + merge per frame     %7.1f us   (world map %s)" % (t(demo_frame, 200), tuple(build.world_map.topdown_map.shape)))
# --- cfg1
W, H, mw, mh = 320, 240, 256, 256
d = torch.empty(1, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
tposes = [torch.tensor([[0.1 * i, 0.05 * i, 0.1 * i]]) for i in range(8)]
builder = dmap.MapBuilder(proj)
print("cfg1: orth_project               %7.1f us" % t(lambda i: proj.orth_project(d, cam_pose=tposes[i % 8]), 2000))
print("cfg1: plot                       %7.1f us" % t(lambda i: builder.plot(depth_map=d, cam_pose=tposes[i % 8])))
local = [builder.plot(depth_map=d, cam_pose=tposes[i]) for i in range(8)]
print("cfg1: merge                      %7.1f us" % t(lambda i: builder.merge(local[i % 8])))
builder.reset()
print("cfg1: step(merge=True)           %7.1f us" % t(lambda i: builder.step(depth_map=d, cam_pose=tposes[i % 8], merge=True)))
