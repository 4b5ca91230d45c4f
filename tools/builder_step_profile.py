"""cProfile of MapBuilder.step(merge=True) at the reference demo's size (B = 1, 320x240 -> 256x256)."""
import cProfile, pstats, os, sys, time
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
for i in range(20): builder.step(depth_map=d, cam_pose=poses[i % 8], merge=True)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for i in range(n): builder.step(depth_map=d, cam_pose=poses[i % 8], merge=True)
torch.cuda.synchronize()
print("step(merge=True): %.1f us/frame" % ((time.perf_counter() - t0) / n * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(n): builder.step(depth_map=d, cam_pose=poses[i % 8], merge=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
