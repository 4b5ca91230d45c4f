"""Premise check for a launch that is not one lockstep round: the cfg2 call (B = 64, 640x480 -> 512x512) from TWO
host-ordered streams at once -- independent batches, own outputs -- against the same number of calls on one stream.
A scatter workgroup fills a CU's LDS, so the second stream's workgroups start where the first's end: out of step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
B, H, W, mh, mw = 64, 480, 640, 512, 512
ROT = 6
g = torch.Generator().manual_seed(1)
depths = [torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda() for _ in range(ROT)]
poses = [torch.empty(B, 3).uniform_(-1, 1, generator=g) for _ in range(8)]
for p in poses: p[:, 2] *= 3.14
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
outs = [(torch.empty(B, 1, mh, mw, device="cuda"), torch.empty(B, 1, mh, mw, dtype=torch.bool, device="cuda")) for _ in range(ROT)]
fos = [(torch.empty(1, mh, mw, device="cuda"), torch.empty(1, mh, mw, dtype=torch.bool, device="cuda")) for _ in range(ROT)]
N = 96
keep = [None] * ROT
def run(streams, fuse):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for j in range(N):
    with torch.cuda.stream(streams[j % len(streams)]):
      if fuse:
        proj.orth_project_and_fuse(depths[j % ROT], cam_pose=poses[j % 8], out=outs[j % ROT], fused_out=fos[j % ROT])
      else:
        keep[j % ROT] = proj.orth_project(depths[j % ROT], cam_pose=poses[j % 8])
  host = time.perf_counter() - t0
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / N * 1e6, host / N * 1e6
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
for fuse in (False, True):
  for name, st in (("one stream", [s1]), ("two streams", [s1, s2]), ("three streams", [s1, s2, s3]), ("one stream", [s1])):
    run(st, fuse)
    us, host = run(st, fuse)
    print("%-28s %-14s %.1f us per call (host %.1f)" % ("orth_project_and_fuse" if fuse else "orth_project", name, us, host))
