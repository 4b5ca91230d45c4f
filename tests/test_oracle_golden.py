"""Pin the CPU oracle (oracle/dm_oracle.c + oracle/oracle.py) against golden
vectors produced by the reference itself (tests/golden/gen_golden.py).

Bar: integer bins / validity / masks bit-exact on every pixel and cell; float
maps bit-exact for max/min (well inside north_star's 1e-5), tolerance 1e-5
relative for the order-dependent sum/mean/prod.
"""
import numpy as np
import pytest

from conftest import assert_masks_equal_away_from_fill, load_golden, load_sweep, project_kwargs


def _check_debug(dbg, g):
  valid = g["valid"]
  np.testing.assert_array_equal(dbg["valid"], valid)
  # int32-saturated in the fixture (gen_golden.py: clamp); oracle is int64
  xb = np.clip(dbg["x_bin"], -2**31, 2**31 - 1)
  zb = np.clip(dbg["z_bin"], -2**31, 2**31 - 1)
  np.testing.assert_array_equal(xb, g["x_bin"])
  np.testing.assert_array_equal(zb, g["z_bin"])
  fin = np.isfinite(g["y"])
  np.testing.assert_array_equal(dbg["y"][fin], g["y"][fin])
  np.testing.assert_array_equal(np.isnan(dbg["y"]), np.isnan(g["y"]))


def _same(a, b):
  """bit-exact float compare, NaN == NaN, -0 == +0."""
  np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name,height", [
    ("g1_height_320x240_256", True),
    ("g2_local_noflip_clip_64x48", False),
    ("g6a_edge_depth_notrunc", True),
    ("g6b_edge_depth_trunc", True),
])
def test_height_maps(oracle, name, height):
  g, cfg = load_golden(name)
  kw = project_kwargs(cfg, oracle.camera_intrinsics)
  out = oracle.orth_project(g["depth"], get_height_map=height, debug=True, **kw)
  _same(out[0], g["topdown"])
  np.testing.assert_array_equal(out[1], g["mask"])
  if height:
    _same(out[2], g["height"])
  _check_debug(out[-1], g)


def test_fill_none_is_zero_canvas(oracle):
  g, cfg = load_golden("g6c_fill_none")
  kw = project_kwargs(cfg, oracle.camera_intrinsics)
  assert kw["fill_value"] is None
  out = oracle.orth_project(g["depth"], **kw)
  _same(out[0], g["topdown"])
  np.testing.assert_array_equal(out[1], g["mask"])


@pytest.mark.parametrize("name", ["g3_semantic_onehot5_64x48", "g3b_semantic_validmap_64x48"])
def test_semantic(oracle, name):
  g, cfg = load_golden(name)
  kw = project_kwargs(cfg, oracle.camera_intrinsics)
  out = oracle.orth_project(g["depth"], value_map=g["value"], valid_map=g.get("valid_map"),
                            get_height_map=True, debug=True, **kw)
  _same(out[0], g["topdown"])
  np.testing.assert_array_equal(out[1], g["mask"])
  _same(np.ascontiguousarray(out[2]), g["height"])
  _check_debug(out[-1], g)


def test_batched_equals_stack_of_single_frames(oracle):
  g, cfg = load_golden("g4_batched4_64x48")
  kw = project_kwargs(cfg, oracle.camera_intrinsics)
  for k in ("cam_pose", "cam_pitch", "cam_height", "width_offset", "height_offset"):
    kw[k] = g[k]
  out = oracle.orth_project(g["depth"], get_height_map=True, debug=True, **kw)
  _same(out[0], g["topdown"])
  np.testing.assert_array_equal(out[1], g["mask"])
  _same(out[2], g["height"])
  _check_debug(out[-1], g)
  # the clamped yaw (|yaw| <= 1e-3) frame really used the identity rotation
  Ry = oracle.rodrigues([0., 1., 0.], g["cam_pose"][:, 2])
  np.testing.assert_array_equal(Ry[3], np.eye(3, dtype=np.float32).reshape(9))


def test_reductions(oracle):
  g, cfg = load_golden("g7_reductions_64x48")
  base = project_kwargs(cfg, oracle.camera_intrinsics)
  for red, fill in (("min", np.inf), ("min", 0.0), ("sum", 0.0), ("sum", 1.0),
                    ("mean", 0.0), ("mean", 2.0), ("prod", 1.0), ("max", 0.0),
                    ("max", 1.0)):
    kw = dict(base, reduction=red, fill_value=fill)
    tag = f"{red}_fill{fill}"
    for prefix, value in (("", g["value"]), ("height_", None)):
      out = oracle.orth_project(g["depth"], value_map=value, **kw)
      want = g[f"{prefix}topdown_{tag}"]
      if red in ("max", "min"):
        _same(out[0], want)
        np.testing.assert_array_equal(out[1], g[f"{prefix}mask_{tag}"])
      else:
        np.testing.assert_allclose(out[0], want, rtol=1e-5, atol=1e-6)
        # the mask may flip only where an order-dependent sum lands within rounding of the fill
        assert_masks_equal_away_from_fill(out[1], g[f"{prefix}mask_{tag}"], want, fill,
                                          exact=(prefix == ""), what=tag)


def test_rodrigues_and_intrinsics(oracle):
  g, _ = load_golden("g10_known_answers")
  Rx = oracle.rodrigues([1., 0., 0.], g["angles"]).reshape(-1, 3, 3)
  Ry = oracle.rodrigues([0., 1., 0.], g["angles"]).reshape(-1, 3, 3)
  # +0 / -0 are the same number; compare values
  np.testing.assert_array_equal(Rx, g["Rx"])
  np.testing.assert_array_equal(Ry, g["Ry"])
  for w, h, hf, vf, cx, cy, fx, fy in g["intrinsics"]:
    got = oracle.camera_intrinsics(int(w), int(h), np.radians(hf),
                                   None if vf < 0 else np.radians(vf))
    assert got == (cx, cy, fx, fy)


def test_fused_equals_max_over_frames(oracle):
  g, cfg = load_golden("g4_batched4_64x48")
  kw = project_kwargs(cfg, oracle.camera_intrinsics)
  for k in ("cam_pose", "cam_pitch", "cam_height"):
    kw[k] = g[k]
  kw["width_offset"] = 32.
  kw["height_offset"] = 32.
  per_frame = oracle.orth_project(g["depth"], **kw)
  fused = oracle.orth_project(g["depth"], fused=True, **kw)
  _same(fused[0], per_frame[0].max(axis=0))
  np.testing.assert_array_equal(fused[1], per_frame[1].any(axis=0))


@pytest.mark.parametrize("k", range(12))
def test_parameter_sweep(oracle, k):
  """g11: 12 seeded configurations run by the reference (pitch of either sign, vfov,
  offsets, flips, local/global, border, truncations, valid maps, min/max, finite fills)."""
  g, cfg = load_sweep()[k]
  kw = project_kwargs(cfg, oracle.camera_intrinsics)
  for name in ("cam_pose", "cam_pitch", "cam_height", "width_offset", "height_offset"):
    kw[name] = g[name]
  top, mask = oracle.orth_project(g["depth"], valid_map=g.get("valid_map"), **kw)
  np.testing.assert_array_equal(mask, g["mask"])
  _same(top, g["topdown"])


# ---------------------------------------------------------------------------
# Rows next to the projector: the oracle's restatements of camera_affine_grid,
# crop_topdown_map and fuse_topdown_maps against the reference-generated fixtures.
# ---------------------------------------------------------------------------
def test_camera_affine_grid_oracle_g8(oracle):
  """maps.py:353-460 on fixture g8: three pose transitions, both flips, a vfov -- bit for bit."""
  g, cfg = load_golden("g8_camera_affine_grid_64x48")
  cx, cy, fx, fy = oracle.camera_intrinsics(cfg["width"], cfg["height"], cfg["hfov"])
  for i in range(3):
    got = oracle.camera_affine_grid(g["depth"], g[f"trans_pose{i}"], cfg["cam_pitch"], cfg["cam_height"],
                                    fx, fy, cx, cy, True)
    np.testing.assert_array_equal(got, g[f"grid{i}"])
  cx, cy, fx, fy = oracle.camera_intrinsics(cfg["width"], cfg["height"], cfg["hfov"], np.radians(50.))
  got = oracle.camera_affine_grid(g["depth"], [0.05, 0.1, 0.02], cfg["cam_pitch"], cfg["cam_height"],
                                  fx, fy, cx, cy, False)
  np.testing.assert_array_equal(got, g["grid_noflip_vfov50"])


def test_camera_affine_grid_oracle_g8b(oracle):
  """The same at BASELINE configs[4]'s frame size (1280 x 960): every 16th row / column."""
  g, cfg = load_golden("g8b_camera_affine_grid_1280x960_sampled")
  h, w, stride = cfg["height"], cfg["width"], int(g["stride"])
  depth = np.random.default_rng(int(g["seed"])).uniform(0.1, 10.0, (1, 1, h, w)).astype(np.float32)
  assert float(depth.astype(np.float64).sum()) == float(g["depth_checksum"])
  cx, cy, fx, fy = oracle.camera_intrinsics(w, h, cfg["hfov"])
  got = oracle.camera_affine_grid(depth, g["trans_pose"], cfg["cam_pitch"], cfg["cam_height"], fx, fy, cx, cy, True)
  np.testing.assert_array_equal(got[:, :, ::stride, ::stride], g["grid_sampled"])


def _crop_offsets(center, cw, ch, woff, hoff, map_height, flip_h):
  """maps.py:2020-2024: the cropped map's offsets."""
  cx, cz = np.float32(center[0]), np.float32(center[1])
  if flip_h:
    cz = np.float32(map_height - 1) - cz
  return (np.float32(np.float32(woff) + np.float32(cw / 2)) - cx,
          np.float32(np.float32(hoff) + np.float32(ch / 2)) - cz)


def test_crop_oracle_g9(oracle):
  """crop_topdown_map (maps.py:1959-2037, utils.py:571-652) on fixture g9: three centre modes x
  three centres (inside, near a corner, out of range), height maps (fill -inf, 'border') and masks."""
  g, cfg = load_golden("g9_topdownmap_queries")
  for mode in ("none", "origin", "camera"):
    src, msk = g[f"{mode}_map"], g[f"{mode}_mask"]
    for ci in range(3):
      center = g[f"{mode}_crop{ci}_center"].astype(np.float32).reshape(1, 2)
      got = oracle.crop_nearest(src, center, 32, 24, -np.inf)
      got_mask = oracle.crop_nearest(msk, center, 32, 24, False)
      np.testing.assert_array_equal(got, g[f"{mode}_crop{ci}_map"])
      np.testing.assert_array_equal(got_mask, g[f"{mode}_crop{ci}_mask"])
      wo, ho = _crop_offsets(center[0], 32, 24, g[f"{mode}_woff"].reshape(-1)[0], g[f"{mode}_hoff"].reshape(-1)[0],
                             cfg["map_height"], True)
      assert wo == g[f"{mode}_crop{ci}_woff"].reshape(-1)[0] and ho == g[f"{mode}_crop{ci}_hoff"].reshape(-1)[0]


def test_crop_oracle_g9b(oracle):
  """The same on a 2048 x 2048 map (BASELINE configs[4]): integral, fractional and out-of-range centres,
  1024 x 1024 and odd crop sizes -- every 8th row / column, the checksums and the shifted offsets."""
  g, cfg = load_golden("g9b_crop_2048_sampled")
  n, stride = int(cfg["map_width"]), int(g["stride"])
  rng = np.random.default_rng(int(g["seed"]))
  top = rng.uniform(-1.0, 3.0, (1, 1, n, n)).astype(np.float32)
  mask = rng.uniform(size=(1, 1, n, n)) > 0.35
  top[~mask] = -np.inf
  assert float(top[np.isfinite(top)].astype(np.float64).sum()) == float(g["top_checksum"])
  for i in range(int(g["ncases"])):
    cw, ch = (int(v) for v in g[f"c{i}_size"])
    center = g[f"c{i}_center"].reshape(1, 2)
    cm = oracle.crop_nearest(top, center, cw, ch, -np.inf)
    ck = oracle.crop_nearest(mask, center, cw, ch, False)
    np.testing.assert_array_equal(cm[:, :, ::stride, ::stride], g[f"c{i}_map"])
    np.testing.assert_array_equal(ck[:, :, ::stride, ::stride], g[f"c{i}_mask"])
    finite = np.isfinite(cm)
    assert [int(finite.sum()), int(ck.sum())] == g[f"c{i}_cells"].tolist()
    assert float(cm[finite].astype(np.float64).sum()) == float(g[f"c{i}_sum"])
    wo, ho = _crop_offsets(center[0], cw, ch, cfg["width_offset"], cfg["height_offset"], n, True)
    assert wo == g[f"c{i}_woff"].reshape(-1)[0] and ho == g[f"c{i}_hoff"].reshape(-1)[0]


def _fuse_sequence(oracle, cfg, step_kw, poses, depths, values, want, semantic):
  """Four MapBuilder.step(merge=True) calls (maps.py:2357-2508) with the oracle: the step's local map
  by oracle.orth_project, the world map by oracle.fuse_topdown_maps(world, local) with the builder's
  projector at the step's pose (merge, maps.py:2471-2508)."""
  intr = oracle.camera_intrinsics(cfg["width"], cfg["height"], cfg["hfov"])
  world = None
  for t in range(4):
    kw = dict(cfg, **step_kw)
    pk = project_kwargs(dict(kw, cam_pose=poses[t]), lambda *a: intr)
    if semantic:
      pk["fill_value"] = 0.
    top, mask, height = oracle.orth_project(depths[t], value_map=values[t] if semantic else None,
                                            get_height_map=True, **pk)
    want["local"](t, top, mask)
    local = dict(height=np.ascontiguousarray(np.broadcast_to(height, top.shape)[0]), mask=mask[0],
                 value=top[0] if semantic else None, woff=kw["width_offset"], hoff=kw["height_offset"],
                 map_res=kw["map_res"], flip_h=True, to_global=bool(kw["to_global"]), cam_pose=poses[t])
    target = dict(map_res=cfg["map_res"], flip_h=True, to_global=bool(cfg["to_global"]), cam_pose=poses[t],
                  fill_value=0. if semantic else -np.inf)
    fused = oracle.fuse_topdown_maps(([world] if world is not None else []) + [local], target)
    assert fused is not None
    want["world"](t, fused)
    world = dict(height=fused["height"], mask=fused["mask"], value=fused["map"] if semantic else None,
                 woff=fused["woff"], hoff=fused["hoff"], map_res=cfg["map_res"], flip_h=True,
                 to_global=bool(cfg["to_global"]), cam_pose=poses[t])


@pytest.mark.parametrize("tag,semantic", [("height", False), ("semantic", True)])
def test_fuse_oracle_g5(oracle, tag, semantic):
  """fuse_topdown_maps (maps.py:2039-2287) on fixture g5: local step maps (res 0.05) fused into a
  growing global world map (res 0.1): size, offsets, content after every step."""
  g, cfg = load_golden(f"g5_builder_merge4_{tag}")
  step_kw = dict(to_global=False, map_res=0.05, width_offset=32., height_offset=0., map_width=64, map_height=64)

  def local(t, top, mask):
    np.testing.assert_array_equal(top, g[f"local_map{t}"])
    np.testing.assert_array_equal(mask, g[f"local_mask{t}"])

  def world(t, f):
    assert [f["map_width"], f["map_height"]] == g[f"world_size{t}"].tolist()
    assert f["woff"] == g[f"world_woff{t}"].reshape(-1)[0] and f["hoff"] == g[f"world_hoff{t}"].reshape(-1)[0]
    np.testing.assert_array_equal(f["mask"][None], g[f"world_mask{t}"])
    np.testing.assert_array_equal(f["map"][None], g[f"world_map{t}"])
    np.testing.assert_array_equal(f["height"][None], g[f"world_height{t}"])

  _fuse_sequence(oracle, cfg, step_kw, g["poses"], [g[f"depth{t}"] for t in range(4)],
                 [g.get(f"value{t}") for t in range(4)], dict(local=local, world=world), semantic)


@pytest.mark.parametrize("seq", [0, 3, 6])
def test_fuse_oracle_g5b(oracle, seq):
  """The random MapBuilder sequences of fixture g5b with center_mode 'none' (the offsets are the
  projector's own: no compute_center_offsets): local and global step maps, height and semantic,
  two resolutions."""
  import os
  from conftest import GOLDEN
  z = np.load(os.path.join(GOLDEN, "g5b_builder_random_sequences.npz"))
  semantic, to_global, mode_i, res100, keep_pose, mw, mh, clip = (int(v) for v in z["meta"][seq])
  assert mode_i == 0 and not keep_pose
  h, w = int(z["height"]), int(z["width"])
  cfg = dict(width=w, height=h, hfov=float(z["hfov"]), cam_pose=[0., 0., 0.], width_offset=mw / 2.,
             height_offset=0. if not to_global else mh / 2., cam_pitch=float(z["pitch"]),
             cam_height=float(z["cam_height"]), map_res=res100 / 100., map_width=mw, map_height=mh,
             trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=clip,
             fill_value=0. if semantic else -np.inf, to_global=bool(to_global))
  k = f"s{seq}_"
  values = [np.eye(3, dtype=np.float32)[z[k + f"labels{t}"].astype(np.int64)].transpose(2, 0, 1).copy()
            if semantic else None for t in range(4)]

  def local(t, top, mask):
    np.testing.assert_array_equal(top, z[k + f"local_map{t}"])
    np.testing.assert_array_equal(np.packbits(mask), z[k + f"local_mask{t}"])

  def world(t, f):
    assert [f["map_width"], f["map_height"]] == z[k + f"world_size{t}"].tolist()
    assert f["woff"] == z[k + f"world_woff{t}"].reshape(-1)[0] and f["hoff"] == z[k + f"world_hoff{t}"].reshape(-1)[0]
    np.testing.assert_array_equal(np.packbits(f["mask"][None]), z[k + f"world_mask{t}"])
    np.testing.assert_array_equal(f["map"][None], z[k + f"world_map{t}"])
    np.testing.assert_array_equal(f["height"][None][:, :1], z[k + f"world_height{t}"])

  _fuse_sequence(oracle, cfg, {}, z[k + "poses"], [z[k + f"depth{t}"] for t in range(4)], values,
                 dict(local=local, world=world), bool(semantic))


def test_point_clouds_oracle_g13(oracle):
  """depth_map_to_point_cloud (maps.py:462-545) and height_map_to_point_cloud (maps.py:547-612) on
  fixture g13: the camera-space cloud of a depth map with 0 / NaN / inf / negative depths (both flips)
  and the cell centres of a height map with per-frame offsets (both flips)."""
  g, _ = load_golden("g13_point_clouds")
  cx, cy, fx, fy = (float(v) for v in g["intr"])
  for tag, flip in (("flip", True), ("noflip", False)):
    pts, (B, c, H, W) = oracle.depth_to_camera_points(g["depth"], cx, cy, fx, fy, flip)
    np.testing.assert_array_equal(pts.reshape(B, c, H, W, 3), g[f"cloud_{tag}"])
    got = oracle._cells_to_points(g["height"], g["woff"], g["hoff"], 0.07, flip)
    np.testing.assert_array_equal(got.reshape(g[f"points_{tag}"].shape), g[f"points_{tag}"])
  with np.errstate(invalid="ignore"):
    ok = (g["depth"] <= np.float32(5.05)) & (g["depth"] >= np.float32(0.15)) & g["valid"]      # maps.py:537-544
  np.testing.assert_array_equal(ok, g["ok_flip"])
  assert g["ok_noflip"].all()


def _close_with_same_specials(got, want, what):
  """Interpolated values: float arithmetic whose order torch's vector kernel does not fix -- within north_star's
  1e-5 (here: 2e-6 absolute on values of a few units), NaN / inf exactly where the reference has them."""
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=what)
  np.testing.assert_array_equal(np.isposinf(got), np.isposinf(want), err_msg=what)
  np.testing.assert_array_equal(np.isneginf(got), np.isneginf(want), err_msg=what)
  fin = np.isfinite(want)
  np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5, atol=2e-6, err_msg=what)


@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
def test_crop_interpolated_oracle_g9c(oracle, mode):
  """crop_topdown_map(mode='bilinear' | 'bicubic') (maps.py:1959-2037, utils.py:571-652) on fixture g9c: a
  height map with empty (-inf) cells, its mask, and a value map with fill None / 0.5, at fractional, integral
  and out-of-range centres -- values within tolerance, masks and NaN / inf patterns equal."""
  g, _ = load_golden("g9c_crop_interpolated")
  for ci in range(3):
    c = g["centers"][ci]
    _close_with_same_specials(oracle.crop_interpolated(g["height"], c, 24, 20, -np.inf, mode), g[f"{mode}_h{ci}_map"],
                              f"{mode} height {ci}")
    np.testing.assert_array_equal(oracle.crop_interpolated(g["mask"], c, 24, 20, False, mode), g[f"{mode}_h{ci}_mask"])
    for tag, fill in (("none", None), ("half", 0.5)):
      _close_with_same_specials(oracle.crop_interpolated(g["value"], c, 24, 20, fill, mode), g[f"{mode}_v{ci}_{tag}"],
                                f"{mode} value {ci} {tag}")
