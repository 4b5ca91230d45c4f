"""The strip path (dm_strip.hip) on the GPU: device-side geometry == the host's, and its maps ==
the CPU oracle, the host-geometry window path and the generic path, on shapes and flag
combinations that exercise owned / shared / never-reached groups, with the split forced so
that small images take it too."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import project_kwargs
from test_strip_geometry import STRIDE, _case, _geometry
from test_window_geometry import _params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dmap():
  import dungeon_maps_amd as dmap
  from dungeon_maps_amd import _native
  _native.lib()
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU (run with -m gpu on an MI355X box)")
  return dmap


def _lib():
  from dungeon_maps_amd import _native
  return _native.lib()


def test_device_geometry_equals_host_geometry(dmap, oracle):
  """k_strip_geometry_dump evaluates the frame geometry the way k_strip_scatter does (one
  wave, lane = strip * 8 + corner); windows, union window and the ok flag must equal the
  host's serial evaluation bit for bit, and so must the cone edges (doubles)."""
  from dungeon_maps_amd import frames
  lib = _lib()
  rng = np.random.default_rng(99)
  compared = 0
  for it in range(60):
    B, H, W, depth, pose, cfg = _case(rng, big_offsets=it % 3 == 0)
    mh = cfg["map_height"]
    intr = oracle.camera_intrinsics(W, H, cfg["hfov"], cfg["vfov"])
    p = _params(B, H, W, cfg, intr)
    table = frames.build_frame_table(B, pose if cfg["to_global"] else None, cfg["cam_pitch"],
                                     cfg["cam_height"], cfg["width_offset"], cfg["height_offset"])
    lib.dm_debug_force_strips(int(rng.integers(1, 9)))
    try:
      P, geom, covers, bound = _geometry(lib, p, table, B, mh, with_covers=False)
      if P <= 0:
        continue
      dev = torch.zeros(B * 336 + 1024, dtype=torch.uint8, device="cuda")
      tab = table.clone()
      if not cfg["to_global"]:        # as the library stages a local map: neutral yaw
        tab[:, 10:19] = torch.tensor([1., 0, 0, 0, 1, 0, 0, 0, 1]); tab[:, 19:21] = 0
      tab_d = tab.cuda()
      got = lib.dm_debug_strip_geometry_dev(ctypes.byref(p), table.data_ptr(), tab_d.data_ptr(),
                                            dev.data_ptr(), dev.numel(), None)
      assert got == P
    finally:
      lib.dm_debug_force_strips(0)
    torch.cuda.synchronize()
    # FrameGeom (336 B): Win16 win[8] (64 B), Win16 U (8 B), Line L[8], R[8] (4 floats each), int ok, inside
    raw = dev.cpu().numpy()[:B * 336].reshape(B, 336)
    wins = raw[:, :64].copy().view(np.int16).reshape(B, 8, 4)
    U = raw[:, 64:72].copy().view(np.int16).reshape(B, 4)
    ok = raw[:, 72 + 2 * 8 * 16:72 + 2 * 8 * 16 + 4].copy().view(np.int32).reshape(B)
    inside = raw[:, 72 + 2 * 8 * 16 + 4:72 + 2 * 8 * 16 + 8].copy().view(np.int32).reshape(B)
    np.testing.assert_array_equal(ok != 0, (geom[:, 0] & 0xff) != 0)
    np.testing.assert_array_equal(inside, geom[:, 0] >> 8)   # (strips whose windows the map did not clip)
    np.testing.assert_array_equal(U.astype(np.int32), geom[:, 4:8])
    np.testing.assert_array_equal(wins.astype(np.int32).reshape(B, 32), geom[:, 8:40])
    compared += B
  assert compared > 40


def _projector(dmap, cfg):
  return dmap.MapProjector(**{k: v for k, v in cfg.items()})


_LISTED = []      # calls per seed that took the value-list path (checked by the test below)


@pytest.mark.parametrize("seed", range(6))
def test_strip_path_equals_oracle_and_other_paths(dmap, oracle, seed):
  lib = _lib()
  rng = np.random.default_rng(4000 + seed)
  ran = listed = 0
  for it in range(14):
    B, H, W, depth, pose, cfg = _case(rng, big_offsets=(it % 5 == 4))
    if W % 4:
      continue
    value = valid = None
    if it % 3 == 1:
      C = int(rng.integers(2, 6))
      value = rng.uniform(-1, 2, size=(B, C, H, W)).astype(np.float32)
      value.reshape(-1)[rng.integers(0, value.size, 200)] = np.nan     # (a NaN never replaces a number)
      cfg["fill_value"] = 0.0 if it % 2 else -np.inf
    if it % 4 == 2:
      valid = rng.uniform(0, 1, size=(B, 1, H, W)) > 0.2
    if it % 7 == 3:
      cfg["reduction"] = "min"; cfg["fill_value"] = np.inf
    kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
    want = oracle.orth_project(depth, value_map=value, valid_map=valid, get_height_map=True, **kw)
    proj = _projector(dmap, cfg)
    d = torch.from_numpy(depth).cuda()
    v = None if value is None else torch.from_numpy(value).cuda()
    m = None if valid is None else torch.from_numpy(valid).cuda()
    strips = int(rng.integers(1, 9))
    lib.dm_debug_force_strips(strips)
    try:
      got = proj.orth_project(d, value_map=v, valid_map=m, cam_pose=pose, get_height_map=True)
      path = lib.dm_debug_last_path()
      fused = proj.orth_project_and_fuse(d, value_map=v, valid_map=m, cam_pose=pose)
    finally:
      lib.dm_debug_force_strips(0)
    torch.cuda.synchronize()
    if path != 2:
      continue                     # (a frame the strip path refuses: covered by the other tests)
    ran += 1
    np.testing.assert_array_equal(got[1].cpu().numpy(), want[1], err_msg=str((cfg, strips)))
    np.testing.assert_array_equal(got[0].cpu().numpy(), want[0], err_msg=str((cfg, strips)))
    np.testing.assert_array_equal(got[2].cpu().numpy(), np.ascontiguousarray(want[2]))
    assert torch.equal(fused[0], got[0]) and torch.equal(fused[1], got[1])
    red = torch.amin if cfg["reduction"] == "min" else torch.amax
    assert torch.equal(fused[2], red(got[0], dim=0))
    lib.dm_debug_force_legacy_window(1)
    try:
      legacy = proj.orth_project(d, value_map=v, valid_map=m, cam_pose=pose)
      assert lib.dm_debug_last_path() != 2
    finally:
      lib.dm_debug_force_legacy_window(0)
    assert torch.equal(legacy[0], got[0]) and torch.equal(legacy[1], got[1])
    if v is not None and v.shape[1] >= 3:
      # value maps of three channels or more took the index pass + value pass; the same call with
      # every channel recomputing its cells must give the same maps
      lib.dm_debug_force_strips(strips); lib.dm_debug_strip_value_list(0)
      try:
        again = proj.orth_project(d, value_map=v, valid_map=m, cam_pose=pose, get_height_map=True)
        assert lib.dm_debug_last_path() == 2
      finally:
        lib.dm_debug_force_strips(0); lib.dm_debug_strip_value_list(1)
      for a_, b_ in zip(again, got):
        assert torch.equal(a_, b_, ) or bool(((a_ == b_) | (a_.isnan() & b_.isnan())).all())
      listed += 1
  assert ran >= 6
  _LISTED.append(listed)


@pytest.mark.parametrize("seed", range(4))
def test_compact_planes_equal_slabs_and_lists_and_the_oracle(dmap, oracle, seed):
  """Round 5: the groups several strips share go through compact planes (frame-wide numbering from the covers,
  k_strip_combine_planes) for calls of at most four strips and two output channels.  The same calls with the
  planes switched off (slabs + lists + k_strip_combine_one, rounds 2-4) and the oracle must agree bit for
  bit: height maps, one- and two-channel value maps, valid maps, min, 1-4 strips; the launch counters say
  which combine kernel ran."""
  lib = _lib()
  rng = np.random.default_rng(5200 + seed)
  ran = 0
  for it in range(16):
    B, H, W, depth, pose, cfg = _case(rng, big_offsets=(it % 5 == 4))
    if W % 4:
      continue
    value = valid = None
    if it % 3 == 1:
      C = int(rng.integers(1, 3))
      value = rng.uniform(-1, 2, size=(B, C, H, W)).astype(np.float32)
      cfg["fill_value"] = 0.0 if it % 2 else -np.inf
    if it % 4 == 2:
      valid = rng.uniform(0, 1, size=(B, 1, H, W)) > 0.2
    if it % 7 == 3:
      cfg["reduction"] = "min"; cfg["fill_value"] = np.inf
    kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
    want = oracle.orth_project(depth, value_map=value, valid_map=valid, get_height_map=True, **kw)
    proj = _projector(dmap, cfg)
    d = torch.from_numpy(depth).cuda()
    v = None if value is None else torch.from_numpy(value).cuda()
    m = None if valid is None else torch.from_numpy(valid).cuda()
    strips = int(rng.integers(1, 5))
    lib.dm_debug_force_strips(strips)
    try:
      outs = []
      for planes in (1, 0):
        lib.dm_debug_planes(planes)
        outs.append(proj.orth_project(d, value_map=v, valid_map=m, cam_pose=pose, get_height_map=True))
        path = lib.dm_debug_last_path()
    finally:
      lib.dm_debug_force_strips(0); lib.dm_debug_planes(-1)
    torch.cuda.synchronize()
    if path != 2:
      continue
    ran += 1
    for a_, b_ in zip(outs[0], outs[1]):
      assert torch.equal(a_, b_) or bool(((a_ == b_) | (a_.isnan() & b_.isnan())).all()), (cfg, strips)
    np.testing.assert_array_equal(outs[0][1].cpu().numpy(), want[1], err_msg=str((cfg, strips)))
    np.testing.assert_array_equal(outs[0][0].cpu().numpy(), want[0], err_msg=str((cfg, strips)))
    np.testing.assert_array_equal(outs[0][2].cpu().numpy(), np.ascontiguousarray(want[2]))
  assert ran >= 5


def test_value_list_path_was_exercised(dmap):
  assert sum(_LISTED) >= 4, _LISTED


def test_cfg2_full_size_strip_path_vs_oracle(dmap, oracle):
  """BASELINE configs[1] at full size through the strip path (the path bench.py measures):
  64 x 640x480 -> 512x512 against the oracle on ALL 64 frames, and == the generic path on all."""
  lib = _lib()
  B, H, W, mh, mw = 64, 480, 640, 512, 512
  g = torch.Generator().manual_seed(1234)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g)
  pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
  pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
             width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
             trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=0, to_global=True,
             fill_value=-np.inf)
  proj = dmap.MapProjector(**cfg)
  d = depth.cuda()
  top, mask, fused, fmask = proj.orth_project_and_fuse(d, cam_pose=pose)
  assert lib.dm_debug_last_path() == 2
  torch.cuda.synchronize()
  want = oracle.orth_project(depth.numpy(), nthreads=16,
                             **dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose.numpy()))
  np.testing.assert_array_equal(mask.cpu().numpy(), want[1])
  np.testing.assert_array_equal(top.cpu().numpy(), want[0])
  # the batch-fused map against the oracle's own fusion of the same frames
  fwant = oracle.orth_project(depth.numpy(), nthreads=16, fused=True,
                              **dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose.numpy()))
  np.testing.assert_array_equal(fmask.cpu().numpy(), fwant[1])
  np.testing.assert_array_equal(fused.cpu().numpy(), fwant[0])
  lib.dm_debug_force_generic_path(1)
  try:
    gtop, gmask, gfused, gfmask = proj.orth_project_and_fuse(d, cam_pose=pose)
  finally:
    lib.dm_debug_force_generic_path(0)
  assert torch.equal(gtop, top) and torch.equal(gmask, mask)
  assert torch.equal(gfused, fused) and torch.equal(gfmask, fmask)
  # every cell of every map is written on every call: poison the outputs' memory first
  for _ in range(3):
    junk = torch.full((B, 1, mh, mw), 123.0, device="cuda"); del junk
    t2, m2 = proj.orth_project(d, cam_pose=pose)
    assert torch.equal(t2, top) and torch.equal(m2, mask)


def _cfg2_like(B=8, H=480, W=640, mh=512, mw=512):
  g = torch.Generator().manual_seed(77)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g)
  poses = []
  for _ in range(3):
    pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
    pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
    poses.append(pose)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
             width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
             trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=0, to_global=True,
             fill_value=-np.inf)
  return depth, poses, cfg


def test_prepared_frames_equal_plain_calls(dmap):
  """MapProjector.prepare: the camera state uploaded once, then projections that only enqueue
  kernels -- bit-equal to orth_project / orth_project_and_fuse, also after update()."""
  lib = _lib()
  depth, poses, cfg = _cfg2_like()
  proj = dmap.MapProjector(**cfg)
  d = depth.cuda()
  lib.dm_debug_force_strips(4)          # (8 frames: the cost model would cut rows too)
  try:
    _prepared_checks(dmap, lib, proj, d, depth, poses, cfg)
  finally:
    lib.dm_debug_force_strips(0)


def _prepared_checks(dmap, lib, proj, d, depth, poses, cfg):
  prep = proj.prepare(depth.shape[0], cam_pose=poses[0])
  for pose in poses:
    prep.update(cam_pose=pose)
    want = proj.orth_project_and_fuse(d, cam_pose=pose)
    assert lib.dm_debug_last_path() == 2
    got = prep.orth_project_and_fuse(d)
    for a, b in zip(got, want):
      assert torch.equal(a, b)
    top, mask, height = prep.orth_project(d, get_height_map=True)
    assert torch.equal(top, want[0]) and torch.equal(mask, want[1]) and height is top
  assert prep.status() == 0
  # value maps + valid maps + height map through the same object type
  C = 3
  g = torch.Generator().manual_seed(5)
  value = torch.empty(depth.shape[0], C, 480, 640).uniform_(-1, 2, generator=g).cuda()
  valid = (torch.empty(depth.shape[0], 1, 480, 640).uniform_(0, 1, generator=g) > 0.3).cuda()
  prep2 = proj.prepare(depth.shape[0], cam_pose=poses[1], value_channels=C, valid_channels=1, fill_value=0.0)
  got = prep2.orth_project(d, value_map=value, valid_map=valid, get_height_map=True)
  want = proj.orth_project(d, value_map=value, valid_map=valid, cam_pose=poses[1], fill_value=0.0,
                           get_height_map=True)
  for a, b in zip(got, want):
    assert torch.equal(a, b)
  # parameters the strip path does not take cannot be prepared
  from dungeon_maps_amd import _native
  with pytest.raises(_native.NativeError):
    proj.prepare(depth.shape[0], cam_pose=poses[0], reduction="sum")
  with pytest.raises(_native.NativeError):
    dmap.MapProjector(**dict(cfg, trunc_depth_max=None)).prepare(depth.shape[0], cam_pose=poses[0])


def test_prepared_single_small_frame_equals_oracle(dmap, oracle):
  """BASELINE configs[0] (B = 1, 320x240 -> 256x256): a plain call of this shape stays on the window
  path, but MapProjector.prepare takes it (column strips of about 40 pixels) -- the prepared
  projection must equal the oracle and the plain call, also for a two-frame batch with a valid map."""
  lib = _lib()
  rng = np.random.default_rng(321)
  for B, with_valid in ((1, False), (2, True)):
    H, W, mh, mw = 240, 320, 256, 256
    depth = rng.uniform(0.1, 8.0, size=(B, 1, H, W)).astype(np.float32)
    valid = (rng.uniform(size=(B, 1, H, W)) > 0.15) if with_valid else None
    pose = np.stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), rng.uniform(-np.pi, np.pi, B)], 1).astype(np.float32)
    cfg = dict(width=W, height=H, hfov=np.radians(70.), vfov=None, cam_pitch=np.radians(-20.), cam_height=0.88,
               width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
               trunc_depth_min=0.15, trunc_depth_max=5.05, trunc_height_max=None, clip_border=0,
               to_global=True, flip_h=True, fill_value=-np.inf, reduction="max")
    proj = dmap.MapProjector(**cfg)
    d = torch.from_numpy(depth).cuda()
    m = None if valid is None else torch.from_numpy(valid).cuda()
    plain = proj.orth_project(d, valid_map=m, cam_pose=pose)
    prep = proj.prepare(B, cam_pose=pose, valid_channels=1 if with_valid else 0)
    got = prep.orth_project(d, valid_map=m)
    assert lib.dm_debug_last_path() == 2 and prep.status() == 0
    kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
    want = oracle.orth_project(depth, valid_map=valid, **kw)
    np.testing.assert_array_equal(got[1].cpu().numpy(), want[1])
    np.testing.assert_array_equal(got[0].cpu().numpy(), want[0])
    assert torch.equal(got[0], plain[0]) and torch.equal(got[1], plain[1])


def test_prepared_projection_far_from_the_origin(dmap, oracle):
  """A prepared batch whose frames lie >= 2^18 cells from the origin: the plan records the QUANTISED
  magnitude and the projection call quantises what the plan holds once more -- the quantisation
  must be idempotent, or dm_orth_project_prepared_f32 refuses the plan dm_frames_prepare_f32 just
  made (round-3 advisor finding: tx = 2242 m at 3 cm gives 393 216 -> 524 288)."""
  lib = _lib()
  rng = np.random.default_rng(2242)
  B, H, W, mh, mw, res = 4, 120, 160, 128, 128, 0.03
  depth = rng.uniform(0.1, 6.0, size=(B, 1, H, W)).astype(np.float32)
  for tx in (1.0, 2242.0, 3500.0, 4300.0, 7000.0):
    pose = np.stack([tx + rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), rng.uniform(-np.pi, np.pi, B)], 1).astype(np.float32)
    woff = float(np.float32(mw / 2. - tx / res))
    cfg = dict(width=W, height=H, hfov=np.radians(70.), vfov=None, cam_pitch=np.radians(-20.), cam_height=0.88,
               width_offset=woff, height_offset=mh / 2., map_res=res, map_width=mw, map_height=mh,
               trunc_depth_min=0.15, trunc_depth_max=5.05, trunc_height_max=None, clip_border=0,
               to_global=True, flip_h=True, fill_value=-np.inf, reduction="max")
    proj = dmap.MapProjector(**cfg)
    d = torch.from_numpy(depth).cuda()
    lib.dm_debug_force_strips(4)
    try:
      prep = proj.prepare(B, cam_pose=pose)
      got = prep.orth_project(d)                    # (raised DM_ERR_UNSUPPORTED for tx = 2242 before the fix)
      assert lib.dm_debug_last_path() == 2 and prep.status() == 0
      plain = proj.orth_project(d, cam_pose=pose)
    finally:
      lib.dm_debug_force_strips(0)
    kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
    want = oracle.orth_project(depth, **kw)
    np.testing.assert_array_equal(got[1].cpu().numpy(), want[1])
    np.testing.assert_array_equal(got[0].cpu().numpy(), want[0])
    assert torch.equal(got[0], plain[0]) and torch.equal(got[1], plain[1])
    assert want[1].any()


def test_prepared_projection_replays_from_a_hip_graph(dmap):
  """dm_orth_project_prepared_f32's launch sequence depends only on the parameters and the
  pointers: captured into a HIP graph it replays bit-equal, and after update() (new poses into
  the same buffers) the SAME graph projects the new poses."""
  depth, poses, cfg = _cfg2_like(B=64)
  proj = dmap.MapProjector(**cfg)
  d = depth.cuda()
  prep = proj.prepare(depth.shape[0], cam_pose=poses[0])
  B, mh, mw = depth.shape[0], cfg["map_height"], cfg["map_width"]
  out = (torch.empty(B, 1, mh, mw, device="cuda"), torch.empty(B, 1, mh, mw, dtype=torch.bool, device="cuda"))
  fout = (torch.empty(1, mh, mw, device="cuda"), torch.empty(1, mh, mw, dtype=torch.bool, device="cuda"))
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    for _ in range(2):                      # warm-up outside the capture (kernel attributes, allocations)
      prep.orth_project_and_fuse(d, out=out, fused_out=fout)
  torch.cuda.current_stream().wait_stream(side)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    prep.orth_project_and_fuse(d, out=out, fused_out=fout)
  for pose in poses:
    prep.update(cam_pose=pose)
    for t in out + fout:
      t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    want = proj.orth_project_and_fuse(d, cam_pose=pose)
    assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])
    assert torch.equal(fout[0], want[2]) and torch.equal(fout[1], want[3])
  assert prep.status() == 0


def test_no_pixel_escapes_its_window_on_the_device(dmap, oracle):
  """dm_debug_count_escapes projects every pixel with the DEVICE's float32 arithmetic and
  counts those that land in the map but outside the window (or per-row cover) of the image
  part that owns them -- what the LDS-windowed paths would drop silently.  Zero, over a sweep
  that includes poses 3e3 ... 3e4 cells from the origin (offsets cancelling the translation)
  and map_res down to 0.004, for the strip path's geometry and the host-geometry windows."""
  from dungeon_maps_amd import frames
  lib = _lib()
  rng = np.random.default_rng(31337)
  landed = strip_cases = window_cases = 0
  for it in range(40):
    B, H, W, depth, pose, cfg = _case(rng, big_offsets=it % 2 == 0)
    mh, mw = cfg["map_height"], cfg["map_width"]
    intr = oracle.camera_intrinsics(W, H, cfg["hfov"], cfg["vfov"])
    p = _params(B, H, W, cfg, intr)
    table = frames.build_frame_table(B, pose if cfg["to_global"] else None, cfg["cam_pitch"],
                                     cfg["cam_height"], cfg["width_offset"], cfg["height_offset"])
    d = torch.from_numpy(depth).cuda()
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    ws = torch.zeros(B * 128 + 256, dtype=torch.uint8, device="cuda")
    # (a) the strip path's device-side geometry
    lib.dm_debug_force_strips(int(rng.integers(1, 9)))
    try:
      P, geom, covers, bound = _geometry(lib, p, table, B, mh)
    finally:
      lib.dm_debug_force_strips(0)
    if P > 0 and bound[0] >= 0 and geom[:, 0].all():
      wins = torch.from_numpy(np.ascontiguousarray(geom[:, 8:8 + 4 * P].reshape(B, P, 4))).cuda()
      both = torch.from_numpy(np.stack(covers, axis=-1).astype(np.uint32).view(np.int32)).cuda()
      rc = lib.dm_debug_count_escapes(ctypes.byref(p), table.data_ptr(), d.data_ptr(), None, wins.data_ptr(),
                                      P, 1, int(geom[0, 2]), H, both.data_ptr(), counts.data_ptr(),
                                      ws.data_ptr(), None)
      assert rc == 0
      n, out_w, out_c = counts.cpu().tolist()
      assert out_w == 0 and out_c == 0, (cfg, n, out_w, out_c)
      landed += n; strip_cases += 1
    # (b) the host-geometry windows of the LDS-windowed path (no depth bands)
    for min_parts in (1, 4):
      parts = (ctypes.c_int32 * 5)()
      n_parts = lib.dm_debug_windows(ctypes.byref(p), table.data_ptr(), min_parts, 1, parts, None, 0)
      wins = np.zeros((B, n_parts, 4), dtype=np.int32)
      lib.dm_debug_windows(ctypes.byref(p), table.data_ptr(), min_parts, 1, parts,
                           wins.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), B * n_parts)
      wd = torch.from_numpy(wins).cuda()
      rc = lib.dm_debug_count_escapes(ctypes.byref(p), table.data_ptr(), d.data_ptr(), None, wd.data_ptr(),
                                      parts[0], parts[1], parts[3], parts[4], None, counts.data_ptr(),
                                      ws.data_ptr(), None)
      assert rc == 0
      n, out_w, _ = counts.cpu().tolist()
      assert out_w == 0, (cfg, list(parts), n, out_w)
      window_cases += 1
  assert landed > 50_000 and strip_cases >= 15 and window_cases >= 60


def test_update_with_another_plan_leaves_the_prepared_batch_untouched(dmap):
  """PreparedProjection.update(): the launch plan is derived on the host first; poses that need
  another plan (here: another camera pitch) raise BEFORE anything is written to the device --
  the object keeps projecting the poses it held, bit-equal to what it gave before."""
  from dungeon_maps_amd import _native
  lib = _lib()
  depth, poses, cfg = _cfg2_like()
  proj = dmap.MapProjector(**cfg)
  d = depth.cuda()
  lib.dm_debug_force_strips(4)
  try:
    prep = proj.prepare(depth.shape[0], cam_pose=poses[0])
    before = prep.orth_project(d)
    plan_before = bytes(prep.plan)
    with pytest.raises(_native.NativeError, match="different launch plan"):
      prep.update(cam_pose=poses[1], cam_pitch=np.radians(-35.))
    assert bytes(prep.plan) == plan_before
    after = prep.orth_project(d)
    assert torch.equal(after[0], before[0]) and torch.equal(after[1], before[1])
    # ... and a matching update still goes through
    prep.update(cam_pose=poses[1])
    want = proj.orth_project(d, cam_pose=poses[1])
    got = prep.orth_project(d)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    with pytest.raises(TypeError):
      prep.update(cam_pose=poses[1], map_res=0.05)
  finally:
    lib.dm_debug_force_strips(0)
  assert prep.status() == 0


def test_frame_records_changed_behind_the_plan_are_reported(dmap):
  """The silent path made loud: a prepared batch whose pose records in device memory no longer
  fit its plan (here: overwritten with a translation of 1e9 m) projects nothing for the frames
  concerned -- their maps hold the fill value -- and raises the sticky status word, which makes
  the NEXT projection call of the process fail without any synchronisation."""
  from dungeon_maps_amd import _native
  lib = _lib()
  depth, poses, cfg = _cfg2_like()
  proj = dmap.MapProjector(**cfg)
  d = depth.cuda()
  lib.dm_debug_force_strips(4)
  try:
    prep = proj.prepare(depth.shape[0], cam_pose=poses[0])
    good = prep.orth_project(d)
    torch.cuda.synchronize()
    _native.check_status()                       # nothing flagged so far
    records = prep.buf[:depth.shape[0] * 48].view(torch.float32).view(-1, 12)
    records[2, 4] = 1e9                          # frame 2: tx
    bad = prep.orth_project(d)
    assert prep.status() == _native.STATUS_FRAME_DID_NOT_FIT
    assert bool(torch.isinf(bad[0][2]).all()) and not bool(bad[1][2].any())     # frame 2: all fill
    for b in (0, 1, 3):
      assert torch.equal(bad[0][b], good[0][b]) and torch.equal(bad[1][b], good[1][b])
    with pytest.raises(_native.NativeError, match="did not fit"):
      proj.orth_project(d, cam_pose=poses[0])
    # the raise cleared the flag: the process carries on
    top, mask = proj.orth_project(d, cam_pose=poses[0])
    assert torch.equal(top, good[0]) and torch.equal(mask, good[1])
    assert prep.status() == 0
  finally:
    lib.dm_debug_force_strips(0)


def test_fifty_calls_in_flight_keep_their_own_poses(dmap, oracle):
  """The reference-signature call (MapProjector.orth_project(depth, cam_pose=...),
  maps.py:1406-1465) with a NEW set of poses on every call, 56 calls enqueued back to back
  without a synchronisation: each call's poses travel in the arguments of its own kernels, so no
  call can see another's.  Every result must equal the same call made alone (checked on the
  generic path, which shares no code with the strips), and three of them the oracle."""
  lib = _lib()
  B, H, W, mh, mw = 6, 480, 640, 512, 512
  g = torch.Generator().manual_seed(2024)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
             width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
             trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=0, to_global=True,
             fill_value=-np.inf)
  proj = dmap.MapProjector(**cfg)
  d = depth.cuda()
  n = 56
  poses = []
  for _ in range(n):
    pose = torch.empty(B, 3).uniform_(-1.5, 1.5, generator=g)
    pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
    poses.append(pose)
  lib.dm_debug_force_strips(4)
  try:
    torch.cuda.synchronize()
    outs = [proj.orth_project_and_fuse(d, cam_pose=p) for p in poses]      # all in flight
    assert lib.dm_debug_last_path() == 2
    torch.cuda.synchronize()
  finally:
    lib.dm_debug_force_strips(0)
  lib.dm_debug_force_generic_path(1)
  try:
    for i, p in enumerate(poses):
      alone = proj.orth_project_and_fuse(d, cam_pose=p)
      torch.cuda.synchronize()
      for a_, b_ in zip(outs[i], alone):
        assert torch.equal(a_, b_), f"call {i} differs from the same call made alone"
  finally:
    lib.dm_debug_force_generic_path(0)
  for i in (0, 27, 55):
    want = oracle.orth_project(depth.numpy(), nthreads=6,
                               **dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=poses[i].numpy()))
    np.testing.assert_array_equal(outs[i][1].cpu().numpy(), want[1])
    np.testing.assert_array_equal(outs[i][0].cpu().numpy(), want[0])


def _strip_info(lib):
  info = (ctypes.c_int32 * 4)()
  lib.dm_debug_last_strip_info(info)
  return list(info)


@pytest.mark.parametrize("groups", [1, 3])
def test_cfg3_geometry_on_the_strip_path_index_and_value_passes(dmap, oracle, groups):
  """BASELINE configs[2] at its geometry ON THE PATH bench.py MEASURES: 40-class one-hot values,
  640x480 -> 512x512, four column strips -- the index pass (every pixel's cell once for all
  channels) and the value pass (one workgroup per strip, channel and frame) -- against the oracle
  on every frame and channel, height map included.  `groups` = 3 caps the slab space so that the
  40 channels go through three channel groups (the route a batch takes whose slabs exceed the
  workspace: the `ch0 > 0` launches of strip_pass)."""
  lib = _lib()
  B, H, W, mh, mw, C = 4, 480, 640, 512, 512, 40
  g = torch.Generator().manual_seed(4040)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g)
  labels = torch.randint(0, C, (B, H, W), generator=g)
  value = torch.nn.functional.one_hot(labels, C).permute(0, 3, 1, 2).float().contiguous()
  pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
  pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
             width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
             trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=0, to_global=True,
             fill_value=0.0)
  proj = dmap.MapProjector(**cfg)
  lib.dm_debug_force_strips(4)
  if groups > 1:
    # slabs of one channel: B frames x 4 strips x (20 K ... 40 K cells) x 4 bytes: room for at most 13 channels
    lib.dm_debug_strip_slab_budget(14 * B * 4 * 20000 * 4 - 1)
  try:
    top, mask, height = proj.orth_project(depth.cuda(), value_map=value.cuda(), cam_pose=pose,
                                          get_height_map=True)
    assert lib.dm_debug_last_path() == 2
    info = _strip_info(lib)
  finally:
    lib.dm_debug_force_strips(0)
    lib.dm_debug_strip_slab_budget(0)
  torch.cuda.synchronize()
  # the value map's pass: one index-pass launch, then value-pass launches per channel group; the
  # height map's pass behind it adds one scatter launch and one group
  assert info[0] == 1, info
  if groups == 1:
    assert info[1] == 2 and info[3] == 2, info
  else:
    assert info[1] >= 1 + 3 and info[3] >= 1 + 3, info
  want = oracle.orth_project(depth.numpy(), value_map=value.numpy(), get_height_map=True, nthreads=4,
                             **dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose.numpy()))
  np.testing.assert_array_equal(mask.cpu().numpy(), want[1])
  np.testing.assert_array_equal(top.cpu().numpy(), want[0])
  np.testing.assert_array_equal(height.cpu().numpy(), np.ascontiguousarray(want[2]))


def _fused_split(lib):
  out = (ctypes.c_int32 * 4)()
  lib.dm_debug_last_fused_split(out)
  return list(out)


@pytest.mark.parametrize("seed", range(4))
def test_fused_projection_on_strips_equals_oracle(dmap, oracle, seed):
  """dm_orth_project_fused_f32 on the strip path (k_strip_fused: one workgroup per strip and GROUP
  of frames, then k_fuse_windows): the fused map of a batch must equal the oracle's, for groups
  of 1, 2, 4 and 8 frames and several strip widths (forced), trajectories and unrelated poses,
  valid maps, a height truncation, min, per-frame camera heights, several depth channels, and on
  top of a running map (`accumulate`)."""
  lib = _lib()
  rng = np.random.default_rng(9100 + seed)
  ran = grouped = 0
  for it in range(10):
    B = int(rng.choice([1, 3, 8, 13, 16]))
    H, W = [(48, 64), (60, 80), (96, 128), (120, 160)][int(rng.integers(4))]
    mh, mw = [(128, 128), (256, 256), (200, 300)][int(rng.integers(3))]
    dc = 2 if it % 5 == 4 else 1
    depth = rng.uniform(0.05, 7.0, size=(B, dc, H, W)).astype(np.float32)
    k = np.arange(B, dtype=np.float32)
    if it % 3 == 0:      # unrelated poses
      pose = np.stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), rng.uniform(-np.pi, np.pi, B)], 1).astype(np.float32)
    else:                # a trajectory
      pose = np.stack([0.3 + 0.02 * k, -0.2 + 0.015 * k, 0.4 + 0.02 * k * (1 if it % 2 else -1)], 1).astype(np.float32)
    is_min = it % 4 == 3
    cfg = dict(width=W, height=H, hfov=float(rng.uniform(0.9, 1.6)), vfov=None,
               cam_pitch=float(rng.uniform(-0.6, 0.1)),
               cam_height=(rng.uniform(0.5, 1.5, B).astype(np.float32) if it % 4 == 1 else 0.88),
               width_offset=mw / 2., height_offset=mh / 2., map_res=float(rng.choice([0.03, 0.05])),
               map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=float(rng.choice([2.5, 5.05])),
               trunc_height_max=None if it % 3 else 0.9, clip_border=int(rng.choice([0, 0, 3])),
               to_global=True, flip_h=bool(it % 5), fill_value=np.inf if is_min else -np.inf,
               reduction="min" if is_min else "max")
    valid = (rng.uniform(size=(B, 1, H, W)) > 0.15) if it % 4 == 2 and dc == 1 else None
    kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
    want, wmask = oracle.orth_project(depth, valid_map=valid, fused=True, **kw)
    proj = _projector(dmap, cfg)
    d = torch.from_numpy(depth).cuda()
    m = None if valid is None else torch.from_numpy(valid).cuda()
    # (unrelated poses: groups of one -- their windows have nothing in common)
    F = 1 if it % 3 == 0 else int(rng.choice([f for f in (1, 2, 4, 8) if f <= B]))
    wp = int(rng.choice([16, 20, 32, 40, 64]))
    lib.dm_debug_force_fused_split(wp, F)
    try:
      fused, fmask = proj.orth_project_fused(d, valid_map=m, cam_pose=pose)
      split = _fused_split(lib)
    finally:
      lib.dm_debug_force_fused_split(0, 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(fmask.cpu().numpy(), wmask, err_msg=str((cfg, F, wp, split)))
    np.testing.assert_array_equal(fused.cpu().numpy(), want, err_msg=str((cfg, F, wp, split)))
    if split[2]:
      ran += 1
      grouped += split[2] > 1
      assert split[0] == (wp + 3) // 4 * 4 and split[2] == F
      # on top of a running map: every cell that is better there stays
      base = torch.from_numpy(rng.uniform(-0.5, 1.5, size=want.shape).astype(np.float32)).cuda()
      run, rmask = proj.orth_project_fused(d, valid_map=m, cam_pose=pose, out=base.clone())
      both = torch.minimum(base, fused) if is_min else torch.maximum(base, fused)
      assert torch.equal(run, both)
  assert ran >= 8 and grouped >= 2, (ran, grouped)


def test_cfg4_runs_on_the_strip_path(dmap, oracle):
  """BASELINE configs[3] per rank (64 frames of one trajectory -> one 1024x1024 map): the call
  must take the strip path (one wave of workgroups), and the map must equal the oracle's with
  the cost model's split and with groups of four frames per workgroup forced
  (tests/test_hip_full_configs.py checks the same call end to end)."""
  lib = _lib()
  B, H, W, mh, mw = 64, 480, 640, 1024, 1024
  g = torch.Generator().manual_seed(77)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g)
  k = torch.arange(B, dtype=torch.float32)
  pose = torch.stack((0.02 * k, 0.01 * k, 0.01 * k), dim=1)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
             width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
             trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=0, to_global=True, fill_value=-np.inf)
  proj = dmap.MapProjector(**cfg)
  d = depth.cuda()
  fused, fmask = proj.orth_project_fused(d, cam_pose=pose)
  split = _fused_split(lib)
  torch.cuda.synchronize()
  assert split[1] > 0 and split[1] * split[3] <= 256, split
  want, wmask = oracle.orth_project(depth.numpy(), fused=True, nthreads=8,
                                    **dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose.numpy()))
  np.testing.assert_array_equal(fmask.cpu().numpy(), wmask)
  np.testing.assert_array_equal(fused.cpu().numpy(), want)
  lib.dm_debug_force_fused_split(40, 4)
  try:
    f4, m4 = proj.orth_project_fused(d, cam_pose=pose)
    assert _fused_split(lib) == [40, 16, 4, 16]
  finally:
    lib.dm_debug_force_fused_split(0, 0)
  assert torch.equal(f4, fused) and torch.equal(m4, fmask)
