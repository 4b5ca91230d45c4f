"""How long one round of the value pass takes: value maps of C channels over B frames with
B * C * 4 workgroups = 1, 2, 4 rounds of the chip's 256 CUs (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
H, W, mh, mw = 480, 640, 512, 512
for B, C in ((16, 4), (32, 4), (64, 4), (64, 8)):
  g = torch.Generator().manual_seed(1)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
  value = torch.nn.functional.one_hot(torch.randint(0, C, (B, H, W), generator=g), C).permute(0, 3, 1, 2).float().contiguous().cuda()
  pose = torch.empty(B, 3).uniform_(-1, 1, generator=g); pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
  proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                           width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                           trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=0.0)
  prep = proj.prepare(B, cam_pose=pose, value_channels=C)
  for _ in range(3):
    prep.orth_project(depth, value_map=value)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10):
    prep.orth_project(depth, value_map=value)
  e1.record(); torch.cuda.synchronize()
  print(f"B={B} C={C}: {B * C * 4} workgroups = {B * C * 4 / 256:.1f} rounds, {e0.elapsed_time(e1) * 100:.1f} us per call (index + value pass + combine)")
