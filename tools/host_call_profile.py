"""Where the host time of one MapProjector.orth_project_and_fuse(depth, cam_pose=...) call goes
(cfg2 shapes; cProfile over calls that never wait for the GPU: a small batch keeps the queue short)."""
import cProfile, pstats, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dungeon_maps_amd as dmap
B, H, W, mh, mw = 64, 480, 640, 512, 512
g = torch.Generator().manual_seed(1)
depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g).cuda()
poses = []
for _ in range(8):
  p = torch.empty(B, 3).uniform_(-1, 1, generator=g); p[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
  poses.append(p)
proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
                         width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
                         trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
outs = [(torch.empty(B, 1, mh, mw, device="cuda"), torch.empty(B, 1, mh, mw, dtype=torch.bool, device="cuda")) for _ in range(2)]
fouts = [(torch.empty(1, mh, mw, device="cuda"), torch.empty(1, mh, mw, dtype=torch.bool, device="cuda")) for _ in range(2)]
_plain = proj.orth_project_and_fuse
proj_call = lambda d, cam_pose, i=[0]: (i.__setitem__(0, i[0] + 1), _plain(d, cam_pose=cam_pose, out=outs[i[0] % 2], fused_out=fouts[i[0] % 2]))[1]
for i in range(20):
  proj_call(depth, cam_pose=poses[i % 8])
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for i in range(n):
  out = proj_call(depth, cam_pose=poses[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/call, with the GPU drained %.1f us/call" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
  out = proj_call(depth, cam_pose=poses[i % 8])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)
