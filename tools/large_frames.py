"""BASELINE configs[4] geometry per GPU (16 frames of 1280x960 into 2048x2048 maps): times
orth_project and camera_affine_grid at a given map resolution.  At map_res 0.01 the frustums
are long thin wedges and the windowed path needs depth bands.
Usage: python tools/large_frames.py [--res 0.03] [--calls 30]   (under rocprofv3: --calls 10)"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=float, default=0.03)
ap.add_argument("--calls", type=int, default=30)
args = ap.parse_args()
B, H, W, mh, mw = 16, 960, 1280, 2048, 2048
g = torch.Generator().manual_seed(5)
depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=args.res, map_width=mw,
                         map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True,
                         fill_value=-np.inf)
shift = torch.tensor([[0.05, 0.1, 0.02]])
for name, fn in (("orth_project", lambda: proj.orth_project(depth, cam_pose=pose)),
                 ("camera_affine_grid", lambda: proj.camera_affine_grid(depth, trans_pose=shift))):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.calls):
    fn()
  torch.cuda.synchronize()
  print(f"{name}: {(time.perf_counter() - t0) / args.calls * 1e6:.1f} us per B={B} batch (map_res {args.res})")
