"""cfg1 (B = 1, 320x240 -> 256x256): DEVICE time per projection by forced number of strips, from a captured HIP graph of
a prepared projection (200 replays back to back: no host in the way)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
lib = _native.lib()
B, H, W, mh, mw = [int(v) for v in os.environ.get("DM_SHAPE", "1,240,320,256,256").split(",")]
g = torch.Generator().manual_seed(1)
d = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
po = torch.empty(B, 3).uniform_(-1, 1, generator=g)
ref = None
for strips, planes in ((0, -1), (8, -1), (4, 1), (4, 0), (2, 1), (1, 1)):
  lib.dm_debug_force_strips(strips); lib.dm_debug_planes(planes)
  proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                           width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                           trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
  try:
    prep = proj.prepare(B, cam_pose=po)
    outs = (torch.empty((B, 1, mh, mw), dtype=torch.float32, device="cuda"), torch.empty((B, 1, mh, mw), dtype=torch.bool, device="cuda"))
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(3): prep.orth_project(d, out=outs)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph): prep.orth_project(d, out=outs)
    for _ in range(5): graph.replay()
    torch.cuda.synchronize()
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): graph.replay()
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = (outs[0].clone(), outs[1].clone())
    same = torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])
    info = (4 * __import__("ctypes").c_int32)(); lib.dm_debug_last_strip_info(info)
    print("strips %d planes %2d: %.2f us per replay  same=%s  path %d launches %s" % (strips, planes, e0.elapsed_time(e1) * 1e3 / n, same, lib.dm_debug_last_path(), list(info)))
    del graph, prep
  except Exception as e:
    print("strips %d planes %d: %s" % (strips, planes, str(e)[:100]))
lib.dm_debug_force_strips(0); lib.dm_debug_planes(-1)
