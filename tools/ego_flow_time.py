"""camera_affine_grid of 16 frames of 1280x960 (BASELINE configs[4] per GPU): HIP-event time per call."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
B, H, W = 16, 960, 1280
g = torch.Generator().manual_seed(1)
d = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=1024., height_offset=1024., map_res=0.03, map_width=2048, map_height=2048,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
tp = torch.tensor([0.05, 0.1, 0.02])
for _ in range(5): out = proj.camera_affine_grid(d, tp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): out = proj.camera_affine_grid(d, tp)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 40
alg = B * H * W * 12
print("camera_affine_grid: %.1f us per call, %.0f GB/s algorithmic, %.3f of 8 TB/s" % (us, alg / us / 1e3, alg / us / 1e3 / 8000))
