"""Host-side logic of the drop-in API on CPU: projector defaults/clone,
Rodrigues matrices and intrinsics against the reference-generated fixture,
frame-table packing, coordinate queries (golden g9), enums."""
import os

import numpy as np
import pytest
import torch

import dungeon_maps_amd as dmap
from dungeon_maps_amd import frames, utils
from conftest import GOLDEN, load_golden


def test_enums_and_known_answers():
  g, _ = load_golden("g10_known_answers")
  assert dmap.Reduction(None) is dmap.Reduction.max
  assert dmap.CenterMode(None) is dmap.CenterMode.none
  got = utils.ravel_index(torch.from_numpy(g["ravel_in"]), (6, 5, 4))
  assert got.tolist() == g["ravel_out"].tolist() == [71, 9]
  assert dmap.get(None, None, 3, 4) == 3 and dmap.get(None, None) is None


def test_rotation_matrices_match_reference():
  g, _ = load_golden("g10_known_answers")
  ang = torch.from_numpy(g["angles"])
  Rx = utils.rotation_matrix([1., 0., 0.], ang).numpy()
  Ry = utils.rotation_matrix([0., 1., 0.], ang).numpy()
  np.testing.assert_array_equal(Rx, g["Rx"])
  np.testing.assert_array_equal(Ry, g["Ry"])
  # rotate() applies out_i = sum_j R[j,i] p_j
  eye = torch.eye(3).view(1, 3, 3).expand(len(ang), 3, 3)
  np.testing.assert_array_equal(utils.rotate(eye, [1., 0., 0.], ang).numpy(), g["Rx"])


def test_intrinsics_match_reference():
  g, _ = load_golden("g10_known_answers")
  for w, h, hf, vf, cx, cy, fx, fy in g["intrinsics"]:
    ci = utils.get_camera_intrinsics(int(w), int(h), np.radians(hf),
                                     None if vf < 0 else np.radians(vf))
    assert (ci.cx, ci.cy, ci.fx, ci.fy) == (cx, cy, fx, fy)


def test_frame_table_closed_form_equals_generic_rodrigues():
  """frames.build_frame_table's closed form == utils.rotation_matrix (the
  op-for-op restatement of reference utils.py:303-327), incl. the clamp."""
  g, _ = load_golden("g10_known_answers")
  rng = np.random.default_rng(0)
  ang = np.concatenate([g["angles"], rng.uniform(-7, 7, 5000).astype(np.float32),
                        np.float32([0.001, -0.001, 0.0010001, 1e-4, 0.0, np.pi / 2, -np.pi / 2])])
  n = len(ang)
  pose = np.zeros((n, 3), dtype=np.float32)
  pose[:, 2] = ang[::-1]
  t = frames.build_frame_table(n, pose, ang, 0.5, 0.0, 0.0)
  Rx = utils.rotation_matrix([1., 0., 0.], torch.from_numpy(ang)).reshape(n, 9)
  Ry = utils.rotation_matrix([0., 1., 0.], torch.from_numpy(ang[::-1].copy())).reshape(n, 9)
  np.testing.assert_array_equal(t[:, 0:9].numpy(), Rx.numpy())
  np.testing.assert_array_equal(t[:, 10:19].numpy(), Ry.numpy())
  k = len(g["angles"])
  np.testing.assert_array_equal(t[:k, 0:9].numpy().reshape(k, 3, 3), g["Rx"])


def test_frame_table_layout_and_broadcast():
  pose = np.array([[0.1, 0.2, 0.3], [0.4, 0.5, -0.6]], dtype=np.float32)
  t = frames.build_frame_table(2, pose, -0.35, [0.8, 0.9], 10.0, [1.0, 2.0])
  assert t.shape == (2, 32) and t.dtype == torch.float32
  Rp = utils.rotation_matrix([1., 0., 0.], torch.tensor([-0.35])).reshape(9)
  assert torch.equal(t[0, 0:9], Rp) and torch.equal(t[1, 0:9], Rp)
  assert t[:, 9].tolist() == pytest.approx([0.8, 0.9])
  Ry = utils.rotation_matrix([0., 1., 0.], torch.tensor([0.3, -0.6])).reshape(2, 9)
  assert torch.equal(t[:, 10:19], Ry)
  assert torch.equal(t[:, 19:21], torch.from_numpy(pose[:, :2]))
  assert t[:, 21].tolist() == [10.0, 10.0] and t[:, 22].tolist() == [1.0, 2.0]
  assert (t[:, 23:] == 0).all()
  with pytest.raises(ValueError):
    frames.build_frame_table(3, pose, 0., 0., 0., 0.)


def test_projector_defaults_clone_and_forwarding():
  p = dmap.MapProjector(width=64, height=48, hfov=1.2)
  assert p.fill_value == dmap.NINF and p.flip_h is True and p.to_global is False
  q = p.clone(map_res=0.05, map_width=10, map_height=12, to_global=True)
  assert (q.map_res, q.map_width, q.map_height, q.to_global) == (0.05, 10, 12, True)
  assert q.width == 64 and q.cam_params == p.cam_params
  with pytest.raises(TypeError):
    p.clone(focal_x=1.0)
  q = q.clone(width_offset=5., height_offset=6.)
  col, row = q.map_quantize([0.1, 0.2], [0.3, 0.4])
  assert col.tolist() == [[7, 9]] and row.tolist() == [[-1, -3]]
  x, z = q.map_dequantize(col, row)
  np.testing.assert_allclose(x.numpy(), [[0.1, 0.2]], atol=0.025)
  # the forwarding methods cache the projector's defaults: a changed field must show at once,
  # an explicit argument wins, an explicit None falls back to the field
  q.width_offset = 6.
  assert q.map_quantize([0.1, 0.2], [0.3, 0.4])[0].tolist() == [[8, 10]]
  assert q.map_quantize([0.1, 0.2], [0.3, 0.4], width_offset=5.)[0].tolist() == [[7, 9]]
  assert q.map_quantize([0.1, 0.2], [0.3, 0.4], width_offset=None)[0].tolist() == [[8, 10]]
  q.map_res = 0.1
  assert q.map_quantize([0.1, 0.2], [0.3, 0.4])[0].tolist() == [[7, 8]]
  with pytest.raises(TypeError):
    q.map_quantize([0.1], [0.3], no_such_argument=1)
  import copy
  r = copy.copy(q)                       # shares the cache dict until an assignment clears it
  assert r.map_quantize([0.1, 0.2], [0.3, 0.4])[0].tolist() == [[7, 8]]
  r.width_offset = 0.
  assert q.map_quantize([0.1, 0.2], [0.3, 0.4])[0].tolist() == [[7, 8]]
  assert r.map_quantize([0.1, 0.2], [0.3, 0.4])[0].tolist() == [[1, 2]]
  assert q.map_quantize([0.1, 0.2], [0.3, 0.4])[0].tolist() == [[7, 8]]


def test_coordinate_queries_match_reference():
  g, cfg = load_golden("g9_topdownmap_queries")
  proj = dmap.MapProjector(**cfg)
  pose = np.array(cfg["cam_pose"], dtype=np.float32)
  build = dmap.MapBuilder(proj)
  pts = torch.tensor([[0.5, 0.0, 1.0], [-1.2, 0.3, 2.2], [0.0, 0.0, 0.0]])
  cds = torch.tensor([[0, 0], [10, 20], [63, 63]], dtype=torch.int64)
  for mode in ("none", "origin", "camera"):
    woff, hoff = build._compute_offsets(cam_pose=pose, center_mode=mode)
    np.testing.assert_array_equal(np.asarray(woff, dtype=np.float32).reshape(-1),
                                  g[f"{mode}_woff"].reshape(-1))
    np.testing.assert_array_equal(np.asarray(hoff, dtype=np.float32).reshape(-1),
                                  g[f"{mode}_hoff"].reshape(-1))
    tm = dmap.TopdownMap(map_projector=proj.clone(cam_pose=pose, width_offset=woff,
                                                  height_offset=hoff))
    np.testing.assert_array_equal(tm.get_camera().numpy(), g[f"{mode}_camera"])
    np.testing.assert_array_equal(tm.get_origin().numpy(), g[f"{mode}_origin"])
    np.testing.assert_array_equal(tm.get_coords(pts, True).numpy(), g[f"{mode}_coords_global"])
    np.testing.assert_array_equal(tm.get_coords(pts, False).numpy(), g[f"{mode}_coords_local"])
    np.testing.assert_array_equal(tm.get_points(cds).numpy(), g[f"{mode}_points"])


def test_topdownmap_container_semantics():
  a = torch.zeros(1, 1, 4, 4)
  tm = dmap.TopdownMap(a, torch.ones_like(a, dtype=torch.bool), a)
  assert tm.is_height_map and tm.height_map is a and tm.map is a and not tm.is_empty
  b = torch.ones(1, 1, 4, 4)
  tm = dmap.TopdownMap(a, None, b)
  assert not tm.is_height_map and tm.height_map is b
  assert dmap.TopdownMap().is_empty


def test_object_api_signatures_match_the_reference():
  """inspect.signature of every public method the reference's MapProjector / TopdownMap /
  MapBuilder declare (fixture g12, generated from maps.py:1253-1749, 1753-1955, 2289-2550):
  same parameter names in the same order, the same ones required."""
  import inspect
  import json
  import dungeon_maps_amd as dmap
  with open(os.path.join(GOLDEN, "g12_api_signatures.json")) as f:
    want = json.load(f)
  problems = []
  for qual, params in sorted(want.items()):
    cls_name, meth = qual.split(".")
    fn = getattr(getattr(dmap, cls_name), meth, None)
    if fn is None:
      problems.append(f"{qual}: missing")
      continue
    got = list(inspect.signature(fn).parameters.values())
    got_names = [p.name for p in got]
    want_names = [p["name"] for p in params]
    if got_names[:len(want_names)] != want_names and got_names != want_names:
      problems.append(f"{qual}: names {got_names} != {want_names}")
      continue
    for g, w in zip(got, params):
      required = g.default is inspect.Parameter.empty and g.kind is not inspect.Parameter.VAR_KEYWORD
      if required != w["required"]:
        problems.append(f"{qual}: {g.name} required={required}, reference {w['required']}")
    extra = got[len(want_names):]
    if any(p.default is inspect.Parameter.empty and p.kind is not inspect.Parameter.VAR_KEYWORD
           for p in extra):
      problems.append(f"{qual}: extra required parameters {[p.name for p in extra]}")
  assert not problems, "\n".join(problems)
  # positional clone(width, height, ...) works as in maps.py:1349-1372
  proj = dmap.MapProjector(64, 48, 1.2, map_res=0.1, map_width=32, map_height=32)
  c = proj.clone(80, 60)
  assert (c.width, c.height, c.map_width) == (80, 60, 32)
  # private plumbing stays out of the public functional signature
  assert "_fuse" not in inspect.signature(dmap.orth_project).parameters


def test_point_cloud_helpers_against_the_reference():
  """functional.depth_map_to_point_cloud / height_map_to_point_cloud (reference maps.py:462-612) on CPU
  tensors against fixture g13, generated by the reference itself: clouds bit for bit, validity equal."""
  import torch
  from conftest import load_golden
  from dungeon_maps_amd import functional as F
  g, _ = load_golden("g13_point_clouds")
  cx, cy, fx, fy = (float(v) for v in g["intr"])
  for tag, kw in (("flip", dict(flip_h=True, trunc_depth_min=0.15, trunc_depth_max=5.05,
                                valid_map=torch.from_numpy(g["valid"]))),
                  ("noflip", dict(flip_h=False, trunc_depth_min=None, trunc_depth_max=None, valid_map=None))):
    cloud, ok = F.depth_map_to_point_cloud(torch.from_numpy(g["depth"]), focal_x=fx, focal_y=fy, center_x=cx,
                                           center_y=cy, **kw)
    np.testing.assert_array_equal(cloud.numpy(), g[f"cloud_{tag}"])
    np.testing.assert_array_equal(ok.numpy(), g[f"ok_{tag}"])
  for tag, flip in (("flip", True), ("noflip", False)):
    pc = F.height_map_to_point_cloud(torch.from_numpy(g["height"]), torch.from_numpy(g["woff"]),
                                     torch.from_numpy(g["hoff"]), 0.07, 20, flip_h=flip)
    np.testing.assert_array_equal(pc.numpy(), g[f"points_{tag}"])


def test_frame_table_takes_one_element_tensors_and_a_flat_pose_array():
  """MapBuilder hands the offsets over as 0-d tensors and the demo its pose as a float32 (3,) array
  (demos/height_map/run.py:104): the same table as from numbers and a (1, 3) tensor."""
  pose = torch.tensor([[0.1, -0.2, 0.3]])
  want = frames.build_frame_table(1, pose, -0.349, 0.88, 128.0, 7.5)
  got = frames.build_frame_table(1, pose, torch.tensor(-0.349), torch.tensor([0.88]), torch.tensor(128.0),
                                 np.array([7.5], dtype=np.float32))
  assert torch.equal(want, got)
  got = frames.build_frame_table(1, np.array([0.1, -0.2, 0.3], dtype=np.float32), -0.349, 0.88, torch.tensor(128.0), 7.5)
  assert torch.equal(want, got)
  # offsets that change from call to call do not grow the cache without bound
  for i in range(600):
    frames.build_frame_table(1, pose, -0.349, 0.88, float(i), 0.0)
  assert len(frames._static_tensors) <= 256
  assert torch.equal(want, frames.build_frame_table(1, pose, -0.349, 0.88, 128.0, 7.5))


def test_fuse_pose_rows_equal_the_rotation_matrix_route():
  """maps._yaw_rows (closed form, what MapBuilder.merge sends to dm_fuse_*) == utils.rotation_matrix about Y
  + the translation, bit for bit and sign for sign (reference maps.py:2039-2069, utils.py:303-327)."""
  from dungeon_maps_amd import maps
  rng = np.random.default_rng(3)
  for trial in range(200):
    b = int(rng.integers(1, 5))
    pose = torch.from_numpy(rng.uniform(-4, 4, (b, 3)).astype(np.float32))
    if trial % 7 == 0:
      pose[:, 2] = torch.from_numpy(rng.uniform(-2e-3, 2e-3, b).astype(np.float32))
    if trial % 11 == 0:
      pose[0, 2] = 0.0
    for inverse in (False, True):
      ang = (-pose[:, 2]).contiguous() if inverse else pose[:, 2].contiguous()
      rot = utils.rotation_matrix(torch.tensor([[0., 1., 0.]]), ang).reshape(-1, 9)
      want = [rot[i].tolist() + ([float(-pose[i, 0]), -0.0, float(-pose[i, 1])] if inverse
                                 else [float(pose[i, 0]), 0.0, float(pose[i, 1])]) for i in range(b)]
      w, g = np.array(want, dtype=np.float32), np.array(maps._yaw_rows(pose, b, inverse), dtype=np.float32)
      np.testing.assert_array_equal(w, g)
      np.testing.assert_array_equal(np.signbit(w), np.signbit(g))


def test_center_offsets_short_cut_equals_the_general_route():
  from dungeon_maps_amd import functional as F
  for woff, hoff in ((300., 0.), (128, 64), (0.5, -3.25)):
    fast = F.compute_center_offsets(np.zeros(3, dtype=np.float32), woff, hoff, 0.03, 600, 600, False, "none")
    slow = F.compute_center_offsets(torch.zeros(3), torch.tensor(float(woff)), torch.tensor(float(hoff)), 0.03, 600,
                                    600, False, F.CenterMode.none)
    for a, b in zip(fast, slow):
      assert a.dtype == b.dtype == torch.float32 and a.shape == b.shape == () and torch.equal(a, b)
  proj = dmap.MapProjector(width=32, height=24, hfov=1.2, width_offset=12.0, height_offset=3.0, map_res=0.1,
                           map_width=64, map_height=64)
  builder = dmap.MapBuilder(proj)
  got = builder._compute_offsets(cam_pose=np.zeros(3, dtype=np.float32), center_mode=dmap.CenterMode.none)
  want = proj.compute_center_offsets(cam_pose=np.zeros(3, dtype=np.float32), center_mode=dmap.CenterMode.none)
  assert all(torch.equal(a, b) and a.dtype == b.dtype and a.shape == b.shape for a, b in zip(got, want))
  got = builder._compute_offsets(cam_pose=np.zeros(3, dtype=np.float32), center_mode=dmap.CenterMode.camera)
  want = proj.compute_center_offsets(cam_pose=np.zeros(3, dtype=np.float32), center_mode=dmap.CenterMode.camera)
  assert all(torch.equal(a, b) for a, b in zip(got, want))


def test_single_frame_table_equals_the_sliced_route():
  """B = 1 takes sin / cos of the whole pose record (no slicing call): the same table as the general route, which
  a two-frame batch with the frame repeated goes through, for 2 000 random yaws and the clamp's neighbourhood."""
  rng = np.random.default_rng(5)
  yaws = np.concatenate([rng.uniform(-7, 7, 2000), [0.0, 1e-3, -1e-3, 1.0000001e-3, 5e-4, np.pi, -np.pi / 2]]).astype(np.float32)
  for yaw in yaws:
    pose = torch.tensor([[0.25, -1.5, float(yaw)]])
    one = frames.build_frame_table(1, pose, -0.349, 0.88, 128.0, 64.0)
    two = frames.build_frame_table(2, pose.repeat(2, 1), -0.349, 0.88, 128.0, 64.0)
    assert torch.equal(one[0], two[0]) and torch.equal(one[0], two[1]), float(yaw)
