"""Host-side geometry of the LDS-windowed path, checked without a GPU: the library bounds the
map window every (frame, image part, depth band) can reach from the poses alone; a pixel that
lands outside its part's window would be lost silently.  Here the oracle projects every pixel
and each valid one must fall inside the window of the part that owns it (dm_debug_windows)."""
import ctypes

import numpy as np
import pytest

import dungeon_maps_amd as dmap
from dungeon_maps_amd import _native, frames, functional as F
from conftest import project_kwargs


def _params(B, H, W, cfg, intr):
  """dm_params as functional._Call builds it; intr = (cx, cy, fx, fy)."""
  key = (B, 1, 0, H, W, int(cfg["map_height"]), int(cfg["map_width"]), int(cfg["clip_border"] or 0),
         bool(cfg["flip_h"]), bool(cfg["to_global"]), F._reduction_code(cfg["reduction"]),
         cfg["trunc_depth_min"], cfg["trunc_depth_max"], cfg["trunc_height_max"], 0,
         float(intr[0]), float(intr[1]), float(intr[2]), float(intr[3]), float(cfg["map_res"]),
         cfg["fill_value"])
  return F._make_params(key)[0]


def _band_edge(dmin, dmax, pd, j, geo):
  """band_edge() of dm_window_geometry.hpp in float32 (fma = exact product, one rounding)."""
  f = np.float32
  dmin, dmax = f(dmin), f(dmax)
  if j <= 0:
    return dmin
  if j >= pd:
    return dmax
  if not geo:
    step = f((dmax - dmin) / f(pd))
    return f(np.float64(j) * np.float64(step) + np.float64(dmin))
  nneg = (3 if pd >= 8 else 1) if (dmin < 0 and dmax > 0) else (pd if dmax <= 0 else 0)
  if j < nneg:
    e = dmin
    for _ in range(j):
      e = f(e * f(0.3))
  elif j == nneg and 0 < nneg < pd:
    e = f(0.0)
  else:
    e = dmax
    for _ in range(j, pd):
      e = f(e * f(0.55))
  return min(max(e, dmin), dmax)


def _band(dmin, dmax, pd, k, geo=0):
  return _band_edge(dmin, dmax, pd, k, geo), _band_edge(dmin, dmax, pd, k + 1, geo)


def _band_mode(cfg):
  """band_mode(): geometric band edges where the depth range reaches beyond the map."""
  dmin, dmax = np.float32(cfg["trunc_depth_min"]), np.float32(cfg["trunc_depth_max"])
  cells = (float(dmax) - float(dmin)) / float(np.float32(cfg["map_res"]))
  return int(dmin < 0 or cells > max(cfg["map_width"], cfg["map_height"]))


def _random_case(rng, fine, open_far=False):
  B = int(rng.integers(1, 5))
  H, W = [(48, 64), (60, 80), (96, 128), (50, 70), (120, 160)][int(rng.integers(5))]
  if fine:
    mh, mw = [(512, 512), (768, 1024), (1024, 640)][int(rng.integers(3))]
    res = float(rng.choice([0.004, 0.006, 0.008, 0.01, 0.0125]))
  else:
    mh, mw = [(64, 64), (96, 128), (128, 96), (256, 256), (300, 200)][int(rng.integers(5))]
    res = float(rng.choice([0.02, 0.03, 0.05, 0.08, 1.0 / 3]))
  depth = rng.uniform(0.05, 8.0, size=(B, 1, H, W)).astype(np.float32)
  if open_far:      # no upper depth bound: far points up to where they leave any map
    far = rng.integers(0, depth.size, depth.size // 4)
    depth.reshape(-1)[far] = rng.uniform(8.0, 400.0, far.size).astype(np.float32)
  # the extremes of the depth range are where a bound would break first
  depth.reshape(-1)[rng.integers(0, depth.size, 64)] = rng.choice(
      np.array([0.0, 0.15, 0.5, 1.5, 2.5, 5.05, 7.0], np.float32), 64)
  pose = np.stack([rng.uniform(-2, 2, B), rng.uniform(-2, 2, B), rng.uniform(-np.pi, np.pi, B)],
                  axis=1).astype(np.float32)
  cfg = dict(width=W, height=H, hfov=float(rng.uniform(0.6, 2.0)),
             vfov=None if rng.integers(2) else float(rng.uniform(0.5, 1.6)),
             cam_pitch=rng.uniform(-0.9, 0.5, size=B).astype(np.float32),
             cam_height=rng.uniform(0.2, 2.0, size=B).astype(np.float32),
             width_offset=float(mw / 2 + rng.uniform(-40, 40)),
             height_offset=float(mh / 2 + rng.uniform(-40, 40)),
             map_res=res, map_width=mw, map_height=mh,
             trunc_depth_min=float(rng.choice([0.0, 0.15, 0.5])),
             trunc_depth_max=None if open_far else float(rng.choice([1.5, 2.5, 5.05, 7.0])),
             trunc_height_max=None if rng.integers(3) else float(rng.uniform(0.2, 1.2)),
             clip_border=int(rng.choice([0, 0, 3, 9])), to_global=bool(rng.integers(2)),
             flip_h=bool(rng.integers(4)), fill_value=-np.inf, reduction="max")
  return B, H, W, depth, pose, cfg


@pytest.mark.parametrize("fine,open_far", [(False, False), (True, False), (False, True), (True, True)])
def test_every_valid_pixel_lands_inside_its_parts_window(oracle, fine, open_far):
  lib = _native.lib()
  rng = np.random.default_rng(20240 + fine + 2 * open_far)
  checked = banded = smaller = 0
  for _ in range(150):
    B, H, W, depth, pose, cfg = _random_case(rng, fine, open_far)
    kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
    *_, dbg = oracle.orth_project(depth, debug=True, **kw)
    intr = oracle.camera_intrinsics(W, H, cfg["hfov"], cfg["vfov"])
    p = _params(B, H, W, cfg, intr)
    table = frames.build_frame_table(B, pose if cfg["to_global"] else None, cfg["cam_pitch"],
                                     cfg["cam_height"], cfg["width_offset"], cfg["height_offset"])
    xb = dbg["x_bin"].reshape(B, H, W); zb = dbg["z_bin"].reshape(B, H, W)
    ok = dbg["valid"].reshape(B, H, W)              # before the in-map test (maps.py:1150-1158)
    ok = ok & (xb >= 0) & (xb < cfg["map_width"]) & (zb >= 0) & (zb < cfg["map_height"])
    for min_parts, pd in ((1, 1), (4, 1)) if open_far else ((1, 1), (4, 1), (1, 2), (2, 4), (1, 8)):
      parts = (ctypes.c_int32 * 5)()
      n = lib.dm_debug_windows(ctypes.byref(p), table.data_ptr(), min_parts, pd, parts, None, 0)
      assert n == parts[0] * parts[1] * parts[2] and parts[2] == pd
      wins = np.zeros((B, n, 4), dtype=np.int32)
      got = lib.dm_debug_windows(ctypes.byref(p), table.data_ptr(), min_parts, pd, parts,
                                 wins.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), B * n)
      assert got == n
      pc, pr, _, wp, hp = list(parts)
      assert pc * wp >= W and pr * hp >= H and wp % 4 == 0
      for k in range(pd):
        if open_far:
          in_band = depth[:, 0] >= np.float32(cfg["trunc_depth_min"])
        else:
          lo, hi = _band(cfg["trunc_depth_min"], cfg["trunc_depth_max"], pd, k, _band_mode(cfg) if pd > 1 else 0)
          in_band = (depth[:, 0] >= lo) & (depth[:, 0] <= hi)
        for iy in range(pr):
          for ix in range(pc):
            rows = slice(iy * hp, min((iy + 1) * hp, H)); cols = slice(ix * wp, min((ix + 1) * wp, W))
            sel = ok[:, rows, cols] & in_band[:, rows, cols]
            x0, z0, w, h = (wins[:, (k * pr + iy) * pc + ix, i][:, None, None] for i in range(4))
            inside = (xb[:, rows, cols] >= x0) & (xb[:, rows, cols] < x0 + w) & \
                     (zb[:, rows, cols] >= z0) & (zb[:, rows, cols] < z0 + h)
            assert not (sel & ~inside).any(), (cfg, parts[:], k, iy, ix)
            checked += int(sel.sum())
      banded += pd > 1
      smaller += int((wins[..., 2] * wins[..., 3] < cfg["map_width"] * cfg["map_height"]).sum())
      assert (wins[..., 0] % 4 == 0).all() and (wins[..., 2] % 4 == 0).all()
      assert (wins[..., 0] >= 0).all() and (wins[..., 0] + wins[..., 2] <= cfg["map_width"]).all()
      assert (wins[..., 1] >= 0).all() and (wins[..., 1] + wins[..., 3] <= cfg["map_height"]).all()
  assert checked > 100_000 and (banded > 0 or open_far)
  assert smaller > 0          # (open far end: the windows are still smaller than the map)


def test_choose_parts_fills_whole_waves():
  """The split is chosen by a cost model: whole waves of 256 workgroups, not more parts than pay."""
  lib = _native.lib()
  base = dict(map_height=512, map_width=512, clip_border=0, flip_h=True, to_global=True,
              reduction="max", trunc_depth_min=0.15, trunc_depth_max=5.05, trunc_height_max=None,
              map_res=0.03, fill_value=-np.inf)
  def split(B, H, W, mh=512, mw=512):
    i = dmap.utils.get_camera_intrinsics(width=W, height=H, hfov=np.radians(70.))
    p = _params(B, H, W, dict(base, map_height=mh, map_width=mw), (i.cx, i.cy, i.fx, i.fy))
    parts = (ctypes.c_int32 * 5)()
    table = frames.build_frame_table(B, None, -0.35, 0.88, mw / 2., mh / 2.)
    lib.dm_debug_windows(ctypes.byref(p), table.data_ptr(), 1, 1, parts, None, 0)
    return parts[0], parts[1]
  assert split(64, 480, 640) == (4, 1)                 # cfg2: 256 workgroups, one per CU
  pc, pr = split(16, 960, 1280, 2048, 2048)
  assert pc * pr == 16                                 # one whole wave, not 20 parts
  pc, pr = split(1, 240, 320, 256, 256)
  assert 4 <= pc * pr <= 16                            # a single small frame is not split 80 ways
  assert split(2560, 480, 640) == (1, 1)               # more frames than CUs: no split
