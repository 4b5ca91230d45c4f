"""Geometry of the strip path, checked without a GPU (dm_debug_strip_geometry returns exactly
what the kernels derive on the device: dm_strip_geometry.hpp compiles for host and device).

The strip path writes a map cell straight from the one strip that can reach it, and merges
only cells two or more strips can reach; a pixel that landed outside its strip's window or
cover would be lost silently.  So: the oracle projects every pixel, and each valid one must
fall inside the window AND the per-row cover of the strip that owns its image column."""
import ctypes

import numpy as np
import pytest

from dungeon_maps_amd import _native, frames
from conftest import project_kwargs
from test_window_geometry import _params

STRIDE = 8 + 4 * 8
ALIGN = 4


def _geometry(lib, p, table, B, mh, with_covers=True):
  geom = np.zeros((B, STRIDE), dtype=np.int32)
  bound = np.zeros(5, dtype=np.int32)
  P = lib.dm_debug_strip_geometry(ctypes.byref(p), table.data_ptr(), geom.ctypes.data, None,
                                  bound.ctypes.data)
  if P <= 0:
    return P, geom, None, bound
  covers = None
  if with_covers:
    both = np.zeros((B, mh, P, 2), dtype=np.uint32)
    assert lib.dm_debug_strip_geometry(ctypes.byref(p), table.data_ptr(), geom.ctypes.data,
                                       both.ctypes.data, bound.ctypes.data) == P
    covers = (both[..., 0].copy(), both[..., 1].copy())
  return P, geom, covers, bound


def _case(rng, big_offsets=False):
  B = int(rng.integers(1, 4))
  H, W = [(48, 64), (60, 80), (96, 128), (120, 160), (240, 320)][int(rng.integers(5))]
  mh, mw = [(64, 64), (96, 128), (128, 96), (256, 256), (300, 200), (512, 512)][int(rng.integers(6))]
  res = float(rng.choice([0.02, 0.03, 0.05, 0.08, 1.0 / 3, 0.004, 0.01]))
  depth = rng.uniform(0.05, 8.0, size=(B, 1, H, W)).astype(np.float32)
  depth.reshape(-1)[rng.integers(0, depth.size, 64)] = rng.choice(
      np.array([0.0, 0.15, 0.5, 1.5, 2.5, 5.05, 7.0], np.float32), 64)
  span = 2.0
  woff = mw / 2 + rng.uniform(-40, 40)
  hoff = mh / 2 + rng.uniform(-40, 40)
  pose = np.stack([rng.uniform(-span, span, B), rng.uniform(-span, span, B),
                   rng.uniform(-np.pi, np.pi, B)], axis=1).astype(np.float32)
  if big_offsets:       # far from the origin, offsets cancelling the translation (|cells| up to 3e4)
    cells = float(rng.choice([3e3, 1e4, 3e4]))
    pose[:, 0] = np.float32(cells * res * rng.choice([-1.0, 1.0]))
    pose[:, 1] = np.float32(cells * res * rng.choice([-1.0, 1.0]))
    woff = mw / 2 - pose[0, 0] / res
    hoff = mh / 2 - pose[0, 1] / res
  pitch = float(rng.uniform(-0.9, 0.4))
  cfg = dict(width=W, height=H, hfov=float(rng.uniform(0.6, 2.0)),
             vfov=None if rng.integers(2) else float(rng.uniform(0.5, 1.6)),
             cam_pitch=pitch, cam_height=float(rng.uniform(0.2, 2.0)),
             width_offset=float(woff), height_offset=float(hoff),
             map_res=res, map_width=mw, map_height=mh,
             trunc_depth_min=float(rng.choice([0.0, 0.15, 0.5])),
             trunc_depth_max=float(rng.choice([1.5, 2.5, 5.05, 7.0])),
             trunc_height_max=None if rng.integers(3) else float(rng.uniform(0.2, 1.2)),
             clip_border=int(rng.choice([0, 0, 3, 9])), to_global=bool(rng.integers(4)),
             flip_h=bool(rng.integers(4)), fill_value=-np.inf, reduction="max")
  return B, H, W, depth, pose, cfg


@pytest.mark.parametrize("big_offsets", [False, True])
def test_every_valid_pixel_lands_inside_its_strips_window_and_cover(oracle, big_offsets):
  lib = _native.lib()
  rng = np.random.default_rng(777 + big_offsets)
  checked = applied = shared_groups = owned_groups = inside_strips = clipped_strips = 0
  for _ in range(120):
    B, H, W, depth, pose, cfg = _case(rng, big_offsets)
    mh, mw = cfg["map_height"], cfg["map_width"]
    kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
    *_, dbg = oracle.orth_project(depth, debug=True, **kw)
    intr = oracle.camera_intrinsics(W, H, cfg["hfov"], cfg["vfov"])
    p = _params(B, H, W, cfg, intr)
    table = frames.build_frame_table(B, pose if cfg["to_global"] else None, cfg["cam_pitch"],
                                     cfg["cam_height"], cfg["width_offset"], cfg["height_offset"])
    lib.dm_debug_force_strips(int(rng.integers(1, 9)))     # any split, not just the cost model's
    try:
      P, geom, covers, bound = _geometry(lib, p, table, B, mh)
    finally:
      lib.dm_debug_force_strips(0)
    if P <= 0 or bound[0] < 0:
      continue                       # the strip path does not take this call
    wp = int(geom[0, 2])
    assert P * wp >= W and wp % 4 == 0 and 1 <= P <= 8
    xb = dbg["x_bin"].reshape(B, H, W); zb = dbg["z_bin"].reshape(B, H, W)
    pre_ok = dbg["valid"].reshape(B, H, W)          # every test but the map's bounds
    ok = pre_ok & (xb >= 0) & (xb < mw) & (zb >= 0) & (zb < mh)
    for b in range(B):
      if not geom[b, 0]:
        continue                     # cone model not applicable to this frame: flagged, not used
      applied += 1
      U = geom[b, 4:8]
      wins = geom[b, 8:8 + 4 * P].reshape(P, 4)
      lo = (covers[0][b] & 0xffff).astype(np.int64); hi = (covers[0][b] >> 16).astype(np.int64)   # (mh, P)
      olo = (covers[1][b] & 0xffff).astype(np.int64); ohi = (covers[1][b] >> 16).astype(np.int64)
      assert ((lo % 4 == 0) & (hi % 4 == 0) & (olo % 4 == 0) & (ohi % 4 == 0)).all()
      # an owned span lies inside its strip's cover and meets no other strip's cover
      has = ohi > olo
      assert (olo[has] >= lo[has]).all() and (ohi[has] <= hi[has]).all()
      for s in range(P):
        for q in range(P):
          if q != s:
            meet = has[:, s] & (hi[:, q] > lo[:, q]) & (lo[:, q] < ohi[:, s]) & (hi[:, q] > olo[:, s])
            assert not meet.any()
      for s in range(P):
        x0, z0, w, h = wins[s]
        assert x0 % 4 == 0 and w % 4 == 0 and x0 >= 0 and x0 + w <= mw and z0 >= 0 and z0 + h <= mh
        if w:       # inside the union window; covers inside the window widened to whole spans of ALIGN cells
          assert U[0] <= x0 and x0 + w <= U[0] + U[2] and U[1] <= z0 and z0 + h <= U[1] + U[3]
          rows = np.arange(mh)
          outside = (rows < z0) | (rows >= z0 + h)
          assert (hi[outside, s] == 0).all()
          live = hi[:, s] > 0
          A = ALIGN        # kSpanAlign of dm_strip_geometry.hpp
          assert (lo[live, s] >= (x0 & ~(A - 1))).all() and (hi[live, s] <= min((x0 + w + A - 1) & ~(A - 1), mw)).all()
          assert (lo[live, s] % A == 0).all() and ((hi[live, s] % A == 0) | (hi[live, s] == mw)).all()
        cols = slice(s * wp, min((s + 1) * wp, W))
        sel = ok[b, :, cols]
        xs, zs = xb[b, :, cols][sel], zb[b, :, cols][sel]
        inside = (xs >= x0) & (xs < x0 + w) & (zs >= z0) & (zs < z0 + h)
        assert inside.all(), (cfg, s, "window")
        in_cover = (xs >= lo[zs, s]) & (xs < hi[zs, s])
        assert in_cover.all(), (cfg, s, "cover", int((~in_cover).sum()))
        checked += int(sel.sum())
        # a strip flagged "inside" (its window was not clipped by the map's borders) runs without
        # the window test: EVERY pixel that passes the other tests must land inside the window
        if (int(geom[b, 0]) >> (8 + s)) & 1:
          pre = pre_ok[b, :, cols]
          xa, za = xb[b, :, cols][pre], zb[b, :, cols][pre]
          assert ((xa >= x0) & (xa < x0 + w) & (za >= z0) & (za < z0 + h)).all(), (cfg, s, "inside")
          inside_strips += 1
        elif w:
          clipped_strips += 1
      # the launch bound holds for these frames
      assert bound[1] == 0 or ((wins[:, 2] * wins[:, 3]).max() <= bound[2] and U[3] <= bound[3]
                               and U[2] * U[3] <= bound[4])
      # ownership statistics: groups in exactly one cover vs. in several
      gx = np.arange(0, mw, 4)
      n = ((gx[None, :, None] >= lo[:, None, :]) & (gx[None, :, None] < hi[:, None, :])).sum(-1)
      no = ((gx[None, :, None] >= olo[:, None, :]) & (gx[None, :, None] < ohi[:, None, :])).sum(-1)
      assert (no <= 1).all() and (n[no == 1] == 1).all()     # owned => in exactly one cover
      owned_groups += int((no == 1).sum()); shared_groups += int(((n >= 1) & (no == 0)).sum())
  assert checked > 200_000 and applied > 60
  assert inside_strips > 20 and clipped_strips > 20
  # most reachable groups have one owner (that is the point of the path)
  assert owned_groups > shared_groups > 0


def test_cfg2_plan_and_bound():
  """BASELINE configs[1]: 4 strips of 160 columns; the launch bound fits LDS for every yaw."""
  lib = _native.lib()
  import dungeon_maps_amd as dmap
  cfg = dict(width=640, height=480, hfov=np.radians(70.), vfov=None, cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=256., height_offset=256., map_res=0.03, map_width=512,
             map_height=512, trunc_depth_min=0.15, trunc_depth_max=5.05, trunc_height_max=None,
             clip_border=0, to_global=True, flip_h=True, fill_value=-np.inf, reduction="max")
  i = dmap.utils.get_camera_intrinsics(width=640, height=480, hfov=np.radians(70.))
  B = 64
  p = _params(B, 480, 640, cfg, (i.cx, i.cy, i.fx, i.fy))
  rng = np.random.default_rng(5)
  pose = np.stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), rng.uniform(-np.pi, np.pi, B)], 1)
  table = frames.build_frame_table(B, pose.astype(np.float32), cfg["cam_pitch"], cfg["cam_height"],
                                   256., 256.)
  P, geom, covers, bound = _geometry(lib, p, table, B, 512)
  assert P == 4 and geom[0, 2] == 160
  slack, fits, cells, rows, ucells = bound
  assert slack == 3 and fits == 1
  areas = geom[:, 8:8 + 16].reshape(B, 4, 4)
  assert (areas[..., 2] * areas[..., 3]).max() <= cells <= 40_000
  assert geom[:, 7].max() <= rows <= 512 and (geom[:, 6] * geom[:, 7]).max() <= ucells
  lo = (covers[0] & 0xffff).astype(np.int64); hi = (covers[0] >> 16).astype(np.int64)
  olo = (covers[1] & 0xffff).astype(np.int64); ohi = (covers[1] >> 16).astype(np.int64)
  gx = np.arange(0, 512, 4)
  n = ((gx[None, None, :, None] >= lo[:, :, None, :]) & (gx[None, None, :, None] < hi[:, :, None, :])).sum(-1)
  no = ((gx[None, None, :, None] >= olo[:, :, None, :]) & (gx[None, None, :, None] < ohi[:, :, None, :])).sum(-1)
  owned, shared = int((no == 1).sum()), int(((n >= 1) & (no == 0)).sum())
  assert int((n == 1).sum()) - owned < 0.02 * owned    # nearly every group in one cover is owned
  assert owned > 2.5 * shared        # ~75 % of the reachable groups go straight to the map


def test_ineligible_calls_are_refused():
  lib = _native.lib()
  import dungeon_maps_amd as dmap
  base = dict(width=64, height=48, hfov=1.2, vfov=None, cam_pitch=-0.3, cam_height=0.9,
              width_offset=32., height_offset=32., map_res=0.1, map_width=64, map_height=64,
              trunc_depth_min=0.15, trunc_depth_max=5.05, trunc_height_max=None, clip_border=0,
              to_global=True, flip_h=True, fill_value=-np.inf, reduction="max")
  i = dmap.utils.get_camera_intrinsics(width=64, height=48, hfov=1.2)
  table = frames.build_frame_table(1, np.zeros((1, 3), np.float32), -0.3, 0.9, 32., 32.)
  def strips(**over):
    cfg = dict(base, **over)
    p = _params(1, 48, 64, cfg, (i.cx, i.cy, i.fx, i.fy))
    geom = np.zeros((1, STRIDE), dtype=np.int32)
    return lib.dm_debug_strip_geometry(ctypes.byref(p), table.data_ptr(), geom.ctypes.data, None, None)
  assert strips() >= 1
  assert strips(trunc_depth_max=None) == 0        # unbounded far end
  assert strips(trunc_depth_min=None) == 0        # negative depths project behind the camera
  assert strips(reduction="sum") == 0             # max / min only
  assert strips(map_width=63) == 0                # 16-byte rows
  # a camera looking up so steeply that some rows do not look forward: cone model refused
  tab2 = frames.build_frame_table(1, np.zeros((1, 3), np.float32), 1.4, 0.9, 32., 32.)
  p = _params(1, 48, 64, dict(base, cam_pitch=1.4, vfov=1.6), dmap_intr(64, 48, 1.2, 1.6))
  geom = np.zeros((1, STRIDE), dtype=np.int32)
  bound = np.zeros(5, dtype=np.int32)
  lib.dm_debug_strip_geometry(ctypes.byref(p), tab2.data_ptr(), geom.ctypes.data, None, bound.ctypes.data)
  assert geom[0, 0] == 0 and bound[1] == 0


def dmap_intr(W, H, hfov, vfov):
  import dungeon_maps_amd as dmap
  i = dmap.utils.get_camera_intrinsics(width=W, height=H, hfov=hfov, vfov=vfov)
  return i.cx, i.cy, i.fx, i.fy
