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
