"""MapBuilder.step(merge=True) at cfg1: where the host time goes (cProfile over 500 steps, top by own and cumulative time),
and plot / merge timed alone."""
import cProfile, pstats, os, sys, time, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
B, H, W, mh, mw = 1, 240, 320, 256, 256
g = torch.Generator().manual_seed(1)
d = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
poses = [torch.tensor([[0.1 * i, 0.05 * i, 0.1 * i]]) for i in range(8)]
builder = dmap.MapBuilder(proj)
def t(fn, n=300):
  for i in range(20): fn(i)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(n): fn(i)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e6
print("step(merge=True)   %.1f us" % t(lambda i: builder.step(depth_map=d, cam_pose=poses[i % 8], merge=True)))
print("plot               %.1f us" % t(lambda i: builder.plot(depth_map=d, cam_pose=poses[i % 8])))
local = [builder.plot(depth_map=d, cam_pose=poses[i]) for i in range(8)]
print("merge              %.1f us" % t(lambda i: builder.merge(local[i % 8])))
print("orth_project       %.1f us" % t(lambda i: proj.orth_project(d, cam_pose=poses[i % 8])))
print("proj.clone         %.1f us" % t(lambda i: proj.clone(cam_pose=poses[i % 8])))
pr = cProfile.Profile()
for i in range(20): builder.step(depth_map=d, cam_pose=poses[i % 8], merge=True)
pr.enable()
for i in range(500): builder.step(depth_map=d, cam_pose=poses[i % 8], merge=True)
pr.disable()
for key in ("tottime", "cumulative"):
  s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28); print(s.getvalue()[:6000])
