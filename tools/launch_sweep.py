"""cfg2 launch sequence and step under the library's measurement switches, on a working set beyond the
Infinity Cache (ROT depth batches / output sets in rotation, as bench.py):
    python tools/launch_sweep.py fill_split=-1,0,2,4,6,8
Prints us per plain orth_project launch sequence (non-temporal fill forced, as the timed steps) and
per orth_project_and_fuse step, 64 calls back to back between one pair of HIP events each."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native
lib = _native.lib()
B, H, W, mh, mw = [int(v) for v in os.environ.get("DM_SHAPE", "64,480,640,512,512").split(",")]
ROT = int(os.environ.get("DM_ROT", "5"))
g = torch.Generator().manual_seed(1234)
depths = [torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda() for _ in range(ROT)]
poses = []
for _ in range(8):
  p = torch.empty(B, 3).uniform_(-1, 1, generator=g); p[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
  poses.append(p)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
outs = [(torch.empty(B, 1, mh, mw, device="cuda"), torch.empty(B, 1, mh, mw, dtype=torch.bool, device="cuda")) for _ in range(ROT)]
fouts = [(torch.empty(1, mh, mw, device="cuda"), torch.empty(1, mh, mw, dtype=torch.bool, device="cuda")) for _ in range(ROT)]

def b2b(fn, n=64):
  for j in range(4): fn(j)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for j in range(n): fn(4 + j)
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / n

keep = [None] * (ROT + 1)
def plain(j):
  keep[j % len(keep)] = None
  keep[j % len(keep)] = proj.orth_project(depths[j % ROT], cam_pose=poses[j % 8])
def step(j):
  proj.orth_project_and_fuse(depths[j % ROT], cam_pose=poses[j % 8], out=outs[j % ROT], fused_out=fouts[j % ROT])

ref = None
for arg in sys.argv[1:] or ["fill_split=4"]:
  name, vals = arg.split("=")
  for v in vals.split(","):
    getattr(lib, "dm_debug_" + name)(int(v))
    if name != "force_nt_fill":
      lib.dm_debug_force_nt_fill(int(os.environ.get("DM_NT", "1")))
    res = []
    for rep in range(3):
      res.append((b2b(plain), b2b(step)))
    if name != "force_nt_fill":
      lib.dm_debug_force_nt_fill(-1)
    top, mask, fused, fmask = proj.orth_project_and_fuse(depths[0], cam_pose=poses[0])
    torch.cuda.synchronize()
    sig = (float(torch.where(torch.isfinite(top), top, torch.zeros_like(top)).double().sum()), int(mask.sum()),
           float(torch.where(torch.isfinite(fused), fused, torch.zeros_like(fused)).double().sum()), int(fmask.sum()))
    if ref is None:
      ref = (top.clone(), mask.clone(), fused.clone(), fmask.clone())
    same = all(torch.equal(a, b) for a, b in zip((top, mask, fused, fmask), ref))
    print(f"{name}={v:>3}: launch {min(r[0] for r in res):6.2f} us (runs {[round(r[0], 1) for r in res]})   "
          f"step {min(r[1] for r in res):6.2f} us (runs {[round(r[1], 1) for r in res]})   same_as_first={same}", flush=True)
