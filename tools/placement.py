"""cfg3 (B = 64, 640x480 + 40 classes -> 512x512): does the call's duration depend on where its buffers lie?
The value maps, the output maps and the masks carved out of ONE allocation at chosen byte offsets from each other;
orth_project_and_fuse(out=...) timed by HIP events (calls back to back), two alternating placements per line."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
B, H, W, M, C = 64, 480, 640, 512, 40
g = torch.Generator(device="cuda").manual_seed(3)
depth = torch.empty(B, 1, H, W, device="cuda").uniform_(0.1, 10.0, generator=g)
pose = torch.empty(B, 3).uniform_(-1, 1); pose[:, 2] *= 3.14
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=M / 2., height_offset=M / 2., map_res=0.03, map_width=M, map_height=M,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=0.0)
nv, no, nm = B * C * H * W * 4, B * C * M * M * 4, B * C * M * M
SLACK = 2 << 30
big = torch.empty(nv + no + nm + 4 * SLACK, dtype=torch.uint8, device="cuda")
base = big.data_ptr()
def carve(off, nbytes, dtype, shape):
  return big[off:off + nbytes].view(dtype).view(shape)
labels = torch.randint(0, C, (B, H, W), device="cuda", generator=g)
fo = (torch.empty(C, M, M, device="cuda"), torch.empty(C, M, M, dtype=torch.bool, device="cuda"))
def run(dv, do, dm_):
  # value at dv, out at nv + SLACK + do, mask at nv + no + 2 SLACK + dm_ (all multiples of 256)
  value = carve(dv, nv, torch.float32, (B, C, H, W))
  value.zero_(); value.scatter_(1, labels.unsqueeze(1), 1.0)
  out = carve(nv + SLACK + do, no, torch.float32, (B, C, M, M))
  mask = carve(nv + no + 2 * SLACK + dm_, nm, torch.bool, (B, C, M, M))
  for _ in range(2):
    proj.orth_project_and_fuse(depth, value_map=value, cam_pose=pose, out=(out, mask), fused_out=fo)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(4):
    proj.orth_project_and_fuse(depth, value_map=value, cam_pose=pose, out=(out, mask), fused_out=fo)
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / 4, (value.data_ptr() - base, out.data_ptr() - base, mask.data_ptr() - base)
print("base address mod 2 MiB: %d KB" % ((base % (2 << 20)) >> 10))
MB = 1 << 20
cases = [(0, 0, 0)]
for k in (2, 6, 16, 34, 64, 130, 256, 514, 1024, 1500):
  cases += [(0, k * MB, 0), (0, 0, k * MB), (k * MB, 0, 0)]
for dv, do, dm_ in cases:
  us, offs = run(dv, do, dm_)
  print("value +%-5d MB  out +%-5d MB  mask +%-5d MB : %.0f us per call" % (dv // MB, do // MB, dm_ // MB, us))
