"""MapBuilder.merge at cfg1: cProfile (own / cumulative time per call) and hand-timed pieces."""
import cProfile, pstats, os, sys, time, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import maps as M, functional as F, _native
W, H, mw, mh = 320, 240, 256, 256
g = torch.Generator().manual_seed(1)
d = torch.empty(1, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
tposes = [torch.tensor([[0.1 * i, 0.05 * i, 0.1 * i]]) for i in range(8)]
builder = dmap.MapBuilder(proj)
local = [builder.plot(depth_map=d, cam_pose=tposes[i]) for i in range(8)]
def t(fn, n=1000):
  for i in range(30): fn(i)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(n): fn(i)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e6
print("merge                       %6.1f us" % t(lambda i: builder.merge(local[i % 8]), 400))
wm = builder.world_map
dev = d.device
print("_fuse_source (world)        %6.1f us" % t(lambda i: M._fuse_source(wm, proj, dev)))
print("_fuse_source (local)        %6.1f us" % t(lambda i: M._fuse_source(local[i % 8], proj, dev)))
print("proj.clone(cam_pose)        %6.1f us" % t(lambda i: proj.clone(cam_pose=tposes[i % 8])))
s1, s2 = M._fuse_source(wm, proj, dev), M._fuse_source(local[0], proj, dev)
print("FuseSrc array of 2          %6.1f us" % t(lambda i: (_native.FuseSrc * 2)(s1[0], s2[0])))
arr = (_native.FuseSrc * 2)(s1[0], s2[0])
lib = _native.lib(); stream = F._stream_ptr(dev)
print("_zeroed_stats               %6.1f us" % t(lambda i: M._zeroed_stats(dev)))
stats = M._zeroed_stats(dev)
print("bbox call                   %6.1f us" % t(lambda i: lib.dm_fuse_bbox_multi_f32(arr, 2, stats.data_ptr(), stream)))
def bbox_sync(i):
  lib.dm_fuse_bbox_multi_f32(arr, 2, stats.data_ptr(), stream)
  return stats.cpu().tolist()
print("bbox call + .cpu().tolist() %6.1f us" % t(bbox_sync))
print("torch.full (1,1,300,300)    %6.1f us" % t(lambda i: torch.full((1, 1, 300, 300), float("-inf"), dtype=torch.float32, device=dev)))
top = torch.full((1, 1, 300, 300), float("-inf"), dtype=torch.float32, device=dev)
print("scatter call                %6.1f us" % t(lambda i: lib.dm_fuse_scatter_multi_f32(arr, 2, 150.0, 150.0, 1, 300, 300, _native.REDUCE_MAX, top.data_ptr(), None, stream)))
print("mask_from_map               %6.1f us" % t(lambda i: F.mask_from_map(top, float("-inf"))))
print("2 x torch.tensor([v])       %6.1f us" % t(lambda i: (torch.tensor([np.float32(1.5)], dtype=torch.float32), torch.tensor([np.float32(2.5)], dtype=torch.float32))))
print("TopdownMap()                %6.1f us" % t(lambda i: M.TopdownMap(topdown_map=top, mask=top, height_map=top, map_projector=proj, is_height_map=True)))
pr = cProfile.Profile(); pr.enable()
for i in range(500): builder.merge(local[i % 8])
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:3500])
