#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE itself.

Runs only in the build container (needs /root/reference); the GPU box never
sees this script's imports -- it only reads the committed ``*.npz`` files.

What is the reference's own code and what is restated here
----------------------------------------------------------
* Everything in ``dungeon_maps/maps.py`` and ``dungeon_maps/utils.py`` is
  imported and executed unmodified from /root/reference (CPU, float32).
* The reference depends on the third-party package ``torch-scatter`` (unpinned,
  /root/reference/setup.py:55-59), which is not installed in this image and is
  not vendored in the reference.  Its only call site on the path is
  ``scatter_method(flat_values, flat_indices, dim=-1, out=flat_canvas)``
  (/root/reference/dungeon_maps/utils.py:475-477).  The module below restates
  torch-scatter's published ``out=`` semantics (reduce *into* ``out``, existing
  values participate, index broadcast to src) with torch-native ops, and is
  registered as ``torch_scatter`` before the reference is imported.
  max/min are exact and order independent, so this is bit-faithful for the
  north-star reduction; sum/mean/prod are order dependent in f32 and are only
  ever compared with a tolerance.

All reference calls use B=1 (the reference crashes for B>1 in utils.rotate,
/root/reference/dungeon_maps/utils.py:303-316) and batched fixtures are a
stack of B=1 results.

Usage:  python tests/golden/gen_golden.py   (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("DM_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------
# torch-scatter restatement (third-party dependency, absent from the image)
# --------------------------------------------------------------------------
def _install_torch_scatter():
  mod = types.ModuleType("torch_scatter")

  def _expand(index, src):
    return index.expand_as(src) if index.shape != src.shape else index

  def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    assert out is not None
    index = _expand(index, src)
    # upstream keeps the old value unless `new > old`: NaN never wins
    s = torch.where(torch.isnan(src), torch.full_like(src, -np.inf), src)
    out.scatter_reduce_(dim, index, s, reduce="amax", include_self=True)
    return out, None

  def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    assert out is not None
    index = _expand(index, src)
    s = torch.where(torch.isnan(src), torch.full_like(src, np.inf), src)
    out.scatter_reduce_(dim, index, s, reduce="amin", include_self=True)
    return out, None

  def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    assert out is not None
    index = _expand(index, src)
    out.scatter_add_(dim, index, src)
    return out

  def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    assert out is not None
    index = _expand(index, src)
    out.scatter_add_(dim, index, src)
    count = torch.zeros_like(out)
    count.scatter_add_(dim, index, torch.ones_like(src))
    count[count < 1] = 1
    out.true_divide_(count)
    return out

  def scatter_mul(src, index, dim=-1, out=None, dim_size=None):
    assert out is not None
    index = _expand(index, src)
    out.scatter_reduce_(dim, index, src, reduce="prod", include_self=True)
    return out

  mod.scatter_max = scatter_max
  mod.scatter_min = scatter_min
  mod.scatter_add = scatter_add
  mod.scatter_mean = scatter_mean
  mod.scatter_mul = scatter_mul
  sys.modules["torch_scatter"] = mod


def _import_reference():
  _install_torch_scatter()
  sys.path.insert(0, REFERENCE)
  import dungeon_maps as dmap  # noqa: E402
  assert dmap.__version__ == "0.0.3a1", dmap.__version__
  return dmap


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def T(a, dtype=torch.float32):
  return torch.tensor(np.asarray(a), dtype=dtype)


def scene_depth(rng, h, w, hfov, pitch, cam_h, far=6.0):
  """Floor plane + a few box walls seen by a pitched pinhole camera."""
  cx, cy = w / 2., h / 2.
  fx = cx / np.tan(hfov / 2.)
  r = np.arange(h, dtype=np.float64)[:, None]
  q = np.arange(w, dtype=np.float64)[None, :]
  yn = ((h - 1) - r - cy) / fx
  xn = (q - cx) / fx + 0 * r
  # local-space y of a unit-depth point: y = cos(p)*yn + sin(p)
  ky = np.cos(pitch) * yn + np.sin(pitch)
  with np.errstate(divide="ignore"):
    floor = np.where(ky < -1e-6, -cam_h / ky, far)
  walls = far * np.ones_like(floor)
  for _ in range(3):
    c0 = rng.integers(0, w - 8)
    c1 = min(w, c0 + rng.integers(4, max(5, w // 3)))
    walls[:, c0:c1] = np.minimum(walls[:, c0:c1], rng.uniform(1.0, far))
  d = np.minimum(floor, walls) + 0 * xn
  d = d + rng.normal(0, 0.002, size=d.shape)
  return np.clip(d, 0.1, far).astype(np.float32)


def run_orth(dmap, proj_kwargs, depth, value=None, valid=None, call_kwargs=None,
             get_height_map=False, intermediates=True):
  """One B=1 reference call. depth (1,h,w); returns dict of numpy arrays."""
  call_kwargs = dict(call_kwargs or {})
  proj = dmap.MapProjector(**proj_kwargs)
  d = T(depth)
  v = None if value is None else T(value)
  m = None if valid is None else torch.tensor(valid, dtype=torch.bool)
  outs = proj.orth_project(
    depth_map=d.clone(), value_map=None if v is None else v.clone(),
    valid_map=None if m is None else m.clone(),
    get_height_map=get_height_map, **call_kwargs)
  res = {"topdown": outs[0].numpy().copy(), "mask": outs[1].numpy().copy()}
  if get_height_map:
    res["height"] = outs[2].contiguous().numpy().copy()
  if intermediates:
    # replay the reference's own sub-steps (maps.py:260-310) to pin the
    # integer bins and the validity of every pixel
    kw = dict(call_kwargs)
    d4 = dmap.utils.to_4D_image(d.clone())
    pc, va = proj.depth_map_to_point_cloud(
      d4, valid_map=None if m is None else dmap.utils.to_4D_image(m.clone()),
      trunc_depth_min=kw.get("trunc_depth_min"),
      trunc_depth_max=kw.get("trunc_depth_max"))
    clip = kw.get("clip_border", proj.clip_border)
    if clip is not None and clip > 0:
      va = dmap.maps._mask_borders(va, clip)
    pitch = T(kw.get("cam_pitch", proj.cam_pitch)).view(-1)
    camh = T(kw.get("cam_height", proj.cam_height)).view(-1)
    pc = dmap.camera_to_local_space(pc, pitch, camh)
    hmax = kw.get("trunc_height_max", proj.trunc_height_max)
    if hmax is not None:
      va = torch.logical_and(pc[..., 1] <= hmax, va)
    to_global = kw.get("to_global", proj.to_global)
    if to_global:
      pose = T(kw.get("cam_pose", proj.cam_pose)).view(-1, 3)
      pc = dmap.local_to_global_space(pc, pose)
    fpc = torch.flatten(pc, -3, -2)
    woff = T(kw.get("width_offset", proj.width_offset)).view(-1)
    hoff = T(kw.get("height_offset", proj.height_offset)).view(-1)
    xb, zb = dmap.map_quantize(
      fpc[..., 0], fpc[..., 2], woff, hoff,
      map_res=kw.get("map_res", proj.map_res),
      map_height=kw.get("map_height", proj.map_height),
      flip_h=kw.get("flip_h", proj.flip_h))
    i32 = lambda t: t.clamp(-2**31, 2**31 - 1).to(torch.int32).numpy().copy()
    res["x_bin"] = i32(xb)
    res["z_bin"] = i32(zb)
    res["valid"] = torch.flatten(va, -2, -1).numpy().copy()
    res["y"] = fpc[..., 1].numpy().copy()
  return res


def save(name, **arrays):
  path = os.path.join(HERE, name + ".npz")
  np.savez_compressed(path, **arrays)
  print(f"{name}: {os.path.getsize(path)/1024:.0f} KiB")


def pack_kwargs(d):
  """Scalars of a projector/call config as 0-d arrays (None -> nan)."""
  out = {}
  for k, v in d.items():
    if v is None:
      out["cfg_" + k] = np.array(np.nan)
    else:
      out["cfg_" + k] = np.asarray(v)
  return out


# --------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------
def g1_g2(dmap):
  rng = np.random.default_rng(101)
  h, w = 240, 320
  depth = rng.uniform(0.1, 10.0, size=(1, h, w)).astype(np.float32)
  cfg = dict(width=w, height=h, hfov=np.radians(70.), cam_pose=[0.37, -0.21, 0.83],
             width_offset=128., height_offset=128., cam_pitch=np.radians(-20.),
             cam_height=0.88, map_res=0.03, map_width=256, map_height=256,
             trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True,
             fill_value=-np.inf)
  r = run_orth(dmap, cfg, depth, get_height_map=True)
  save("g1_height_320x240_256", depth=depth, **pack_kwargs(cfg), **r)

  h, w = 48, 64
  depth = rng.uniform(0.1, 6.0, size=(1, h, w)).astype(np.float32)
  cfg = dict(width=w, height=h, hfov=np.radians(79.), vfov=np.radians(55.),
             cam_pose=[0.5, 0.25, -2.1], width_offset=20., height_offset=3.5,
             cam_pitch=np.radians(-31.), cam_height=1.25, map_res=0.1,
             map_width=64, map_height=48, trunc_depth_min=0.3,
             trunc_depth_max=5.5, trunc_height_max=1.0, clip_border=5,
             to_global=False, flip_h=False, fill_value=-np.inf)
  r = run_orth(dmap, cfg, depth, get_height_map=False)
  save("g2_local_noflip_clip_64x48", depth=depth, **pack_kwargs(cfg), **r)


def g3(dmap):
  rng = np.random.default_rng(303)
  h, w, C = 48, 64, 5
  depth = scene_depth(rng, h, w, np.radians(70.), np.radians(-20.), 0.88)[None]
  labels = rng.integers(0, C, size=(h, w))
  value = np.eye(C, dtype=np.float32)[labels].transpose(2, 0, 1).copy()
  cfg = dict(width=w, height=h, hfov=np.radians(70.), cam_pose=[-0.4, 0.3, 0.6],
             width_offset=32., height_offset=32., cam_pitch=np.radians(-20.),
             cam_height=0.88, map_res=0.1, map_width=64, map_height=64,
             trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True,
             fill_value=0.)
  r = run_orth(dmap, cfg, depth, value=value, get_height_map=True)
  save("g3_semantic_onehot5_64x48", depth=depth, value=value,
       **pack_kwargs(cfg), **r)
  # valid_map variant (random holes)
  valid = rng.uniform(size=(1, h, w)) > 0.3
  r = run_orth(dmap, cfg, depth, value=value, valid=valid, get_height_map=True)
  save("g3b_semantic_validmap_64x48", depth=depth, value=value, valid_map=valid,
       **pack_kwargs(cfg), **r)


def g4(dmap):
  rng = np.random.default_rng(404)
  B, h, w = 4, 48, 64
  depth = np.stack([
    scene_depth(rng, h, w, np.radians(70.), np.radians(-15. - 5 * b), 0.8 + 0.1 * b)
    for b in range(B)])[:, None]  # (B,1,h,w)
  depth[1] = rng.uniform(0.1, 8.0, size=(1, h, w)).astype(np.float32)
  pose = rng.uniform(-1, 1, size=(B, 3)).astype(np.float32)
  pose[:, 2] = rng.uniform(-np.pi, np.pi, size=B)
  pose[3, 2] = 0.0005  # |yaw| <= ANGLE_EPS -> clamped to zero (utils.py:323-324)
  pitch = np.radians(np.array([-15., -20., -25., 0.0], dtype=np.float32))
  camh = np.array([0.8, 0.9, 1.0, 1.1], dtype=np.float32)
  woff = np.array([32., 30.5, 40., 28.25], dtype=np.float32)
  hoff = np.array([32., 10., 33.75, 36.], dtype=np.float32)
  cfg = dict(width=w, height=h, hfov=np.radians(70.), map_res=0.1, map_width=64,
             map_height=64, trunc_depth_min=0.15, trunc_depth_max=5.05,
             to_global=True, fill_value=-np.inf)
  outs = []
  for b in range(B):
    call = dict(cam_pose=pose[b], cam_pitch=pitch[b:b + 1], cam_height=camh[b:b + 1],
                width_offset=woff[b:b + 1], height_offset=hoff[b:b + 1])
    outs.append(run_orth(dmap, cfg, depth[b], call_kwargs=call, get_height_map=True))
  stacked = {k: np.concatenate([o[k] for o in outs], axis=0) for k in outs[0]}
  save("g4_batched4_64x48", depth=depth, cam_pose=pose, cam_pitch=pitch,
       cam_height=camh, width_offset=woff, height_offset=hoff,
       **pack_kwargs(cfg), **stacked)


def g5(dmap):
  """MapBuilder.step(merge=True) x4 (maps.py:2357-2508): world map after each step."""
  rng = np.random.default_rng(505)
  h, w = 48, 64
  hfov, pitch, camh = np.radians(70.), np.radians(-20.), 0.88
  cfg = dict(width=w, height=h, hfov=hfov, cam_pose=[0., 0., 0.], width_offset=0.,
             height_offset=0., cam_pitch=pitch, cam_height=camh, map_res=0.1,
             map_width=64, map_height=64, trunc_depth_min=0.15,
             trunc_depth_max=5.05, clip_border=2, fill_value=-np.inf,
             to_global=True)
  for tag, semantic in (("height", False), ("semantic", True)):
    proj = dmap.MapProjector(**cfg)
    if semantic:
      proj = proj.clone(fill_value=0.)
      proj.fill_value = 0.
    build = dmap.MapBuilder(map_projector=proj)
    build.reset()
    arrays = {}
    poses = np.array([[0., 0., 0.], [0.3, 0.2, 0.4], [0.5, 0.6, 1.1], [-0.4, 0.9, 2.0]],
                     dtype=np.float32)
    C = 3
    for t in range(4):
      depth = scene_depth(rng, h, w, hfov, pitch, camh)[None]
      value = None
      if semantic:
        labels = rng.integers(0, C, size=(h, w))
        value = np.eye(C, dtype=np.float32)[labels].transpose(2, 0, 1).copy()
      local = build.step(
        depth_map=T(depth), value_map=None if value is None else T(value),
        cam_pose=poses[t].copy(), to_global=False, map_res=0.05,
        width_offset=32., height_offset=0., map_width=64, map_height=64,
        center_mode=dmap.CenterMode.none, merge=True)
      wm = build.world_map
      arrays[f"depth{t}"] = depth
      if semantic:
        arrays[f"value{t}"] = value
      arrays[f"local_map{t}"] = local.topdown_map.numpy().copy()
      arrays[f"local_mask{t}"] = local.mask.numpy().copy()
      arrays[f"local_height{t}"] = local.height_map.contiguous().numpy().copy()
      arrays[f"world_map{t}"] = wm.topdown_map.numpy().copy()
      arrays[f"world_mask{t}"] = wm.mask.numpy().copy()
      arrays[f"world_height{t}"] = wm.height_map.contiguous().numpy().copy()
      arrays[f"world_woff{t}"] = np.asarray(wm.proj.width_offset, dtype=np.float32)
      arrays[f"world_hoff{t}"] = np.asarray(wm.proj.height_offset, dtype=np.float32)
      arrays[f"world_size{t}"] = np.array([wm.proj.map_width, wm.proj.map_height])
    save(f"g5_builder_merge4_{tag}", poses=poses, **pack_kwargs(cfg), **arrays)


def g6(dmap):
  """Edge depths: NaN, +-inf, 0, negative; with and without truncation."""
  h, w = 8, 16
  rng = np.random.default_rng(606)
  depth = rng.uniform(0.5, 3.0, size=(1, h, w)).astype(np.float32)
  depth[0, 0, :6] = [np.nan, np.inf, -np.inf, 0.0, -1.5, -0.0]
  depth[0, 3, 5:9] = [1e30, -1e30, 1e-30, 3.4e38]
  depth[0, 7, 8] = np.nan
  base = dict(width=w, height=h, hfov=np.radians(90.), cam_pose=[0.1, -0.2, 0.3],
              width_offset=16., height_offset=16., cam_pitch=np.radians(-10.),
              cam_height=1.0, map_res=0.2, map_width=32, map_height=32,
              to_global=True, fill_value=-np.inf)
  r = run_orth(dmap, base, depth, get_height_map=True)
  save("g6a_edge_depth_notrunc", depth=depth, **pack_kwargs(base), **r)
  cfg = dict(base, trunc_depth_min=0.15, trunc_depth_max=5.05)
  r = run_orth(dmap, cfg, depth, get_height_map=True)
  save("g6b_edge_depth_trunc", depth=depth, **pack_kwargs(cfg), **r)
  # raw functional default: fill_value None -> zeros canvas (maps.py:320)
  cfg = dict(base, fill_value=None)
  proj = dmap.MapProjector(**cfg)
  proj.fill_value = None
  outs = proj.orth_project(T(depth), fill_value=None, get_height_map=False)
  # `get(None, self.fill_value)` -> None only because proj.fill_value is None
  save("g6c_fill_none", depth=depth, **pack_kwargs(cfg),
       topdown=outs[0].numpy().copy(), mask=outs[1].numpy().copy())


def g7(dmap):
  """Other reductions (order dependent for sum/mean/prod: tolerance only)."""
  rng = np.random.default_rng(707)
  h, w = 48, 64
  depth = scene_depth(rng, h, w, np.radians(70.), np.radians(-20.), 0.88)[None]
  value = rng.uniform(0.5, 1.5, size=(2, h, w)).astype(np.float32)
  cfg = dict(width=w, height=h, hfov=np.radians(70.), cam_pose=[0.2, 0.1, -0.3],
             width_offset=16., height_offset=16., cam_pitch=np.radians(-20.),
             cam_height=0.88, map_res=0.2, map_width=32, map_height=32,
             trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True)
  arrays = dict(depth=depth, value=value)
  for red, fill in (("min", np.inf), ("min", 0.0), ("sum", 0.0), ("sum", 1.0),
                    ("mean", 0.0), ("mean", 2.0), ("prod", 1.0), ("max", 0.0),
                    ("max", 1.0)):
    c = dict(cfg, fill_value=fill, reduction=red)
    r = run_orth(dmap, c, depth, value=value, get_height_map=False,
                 intermediates=False)
    tag = f"{red}_fill{fill}"
    arrays[f"topdown_{tag}"] = r["topdown"]
    arrays[f"mask_{tag}"] = r["mask"]
    r = run_orth(dmap, c, depth, get_height_map=False, intermediates=False)
    arrays[f"height_topdown_{tag}"] = r["topdown"]
    arrays[f"height_mask_{tag}"] = r["mask"]
  save("g7_reductions_64x48", **pack_kwargs(cfg), **arrays)


def g8(dmap):
  """camera_affine_grid (maps.py:353-460)."""
  rng = np.random.default_rng(808)
  h, w = 48, 64
  depth = scene_depth(rng, h, w, np.radians(70.), np.radians(-20.), 0.88)[None]
  depth[0, 0, 0] = 0.0
  cfg = dict(width=w, height=h, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88)
  proj = dmap.MapProjector(**cfg)
  arrays = dict(depth=depth)
  for i, tp in enumerate(([0., 0., 0.], [0.05, 0.1, 0.02], [-0.2, 0.15, -0.4])):
    g = proj.camera_affine_grid(T(depth), T(tp))
    arrays[f"trans_pose{i}"] = np.asarray(tp, dtype=np.float32)
    arrays[f"grid{i}"] = g.numpy().copy()
  proj2 = dmap.MapProjector(**dict(cfg, flip_h=False, vfov=np.radians(50.)))
  g = proj2.camera_affine_grid(T(depth), T([0.05, 0.1, 0.02]))
  arrays["grid_noflip_vfov50"] = g.numpy().copy()
  save("g8_camera_affine_grid_64x48", **pack_kwargs(cfg), **arrays)


def g8b(dmap):
  """camera_affine_grid at BASELINE configs[4]'s frame size (1280x960).  The depth map is
  regenerated from the seed by the test (numpy PCG64 uniform); the fixture keeps the
  reference's grid at every 16th row / column (the op is per pixel)."""
  h, w, seed, stride = 960, 1280, 809, 16
  depth = np.random.default_rng(seed).uniform(0.1, 10.0, (1, 1, h, w)).astype(np.float32)
  cfg = dict(width=w, height=h, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88)
  proj = dmap.MapProjector(**cfg)
  tp = np.asarray([0.05, 0.1, 0.02], dtype=np.float32)
  g = proj.camera_affine_grid(T(depth), T(tp)).numpy()
  save("g8b_camera_affine_grid_1280x960_sampled", **pack_kwargs(cfg), seed=np.int64(seed),
       stride=np.int64(stride), trans_pose=tp, depth_checksum=np.float64(depth.astype(np.float64).sum()),
       grid_sampled=g[:, :, ::stride, ::stride].copy())


def g9(dmap):
  """TopdownMap.select / get_camera / get_origin / get_coords / get_points and
  compute_center_offsets (maps.py:1824-1949, 1959-2037, 1175-1248)."""
  rng = np.random.default_rng(909)
  h, w = 48, 64
  depth = scene_depth(rng, h, w, np.radians(70.), np.radians(-20.), 0.88)[None]
  cfg = dict(width=w, height=h, hfov=np.radians(70.), cam_pose=[0.3, -0.2, 0.5],
             width_offset=32., height_offset=32., cam_pitch=np.radians(-20.),
             cam_height=0.88, map_res=0.1, map_width=64, map_height=64,
             trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True,
             fill_value=-np.inf)
  arrays = dict(depth=depth)
  proj = dmap.MapProjector(**cfg)
  build = dmap.MapBuilder(map_projector=proj)
  for mode in ("none", "origin", "camera"):
    tm = build.plot(T(depth), cam_pose=np.array(cfg["cam_pose"], dtype=np.float32),
                    center_mode=mode)
    arrays[f"{mode}_map"] = tm.topdown_map.numpy().copy()
    arrays[f"{mode}_mask"] = tm.mask.numpy().copy()
    arrays[f"{mode}_woff"] = np.asarray(tm.proj.width_offset, dtype=np.float32)
    arrays[f"{mode}_hoff"] = np.asarray(tm.proj.height_offset, dtype=np.float32)
    arrays[f"{mode}_camera"] = tm.get_camera().numpy().copy()
    arrays[f"{mode}_origin"] = tm.get_origin().numpy().copy()
    pts = T([[0.5, 0.0, 1.0], [-1.2, 0.3, 2.2], [0.0, 0.0, 0.0]])
    arrays[f"{mode}_coords_global"] = tm.get_coords(pts.clone(), is_global=True).numpy().copy()
    arrays[f"{mode}_coords_local"] = tm.get_coords(pts.clone(), is_global=False).numpy().copy()
    cds = torch.tensor([[0, 0], [10, 20], [63, 63]], dtype=torch.int64)
    arrays[f"{mode}_points"] = tm.get_points(cds).numpy().copy()
    for ci, center in enumerate(([32, 32], [5, 60], [70, -3])):
      c = torch.tensor([center], dtype=torch.int64)
      crop = tm.select(c.clone(), 32, 24)
      arrays[f"{mode}_crop{ci}_center"] = np.asarray(center)
      arrays[f"{mode}_crop{ci}_map"] = crop.topdown_map.numpy().copy()
      arrays[f"{mode}_crop{ci}_mask"] = crop.mask.numpy().copy()
      arrays[f"{mode}_crop{ci}_woff"] = np.asarray(crop.proj.width_offset, dtype=np.float32)
      arrays[f"{mode}_crop{ci}_hoff"] = np.asarray(crop.proj.height_offset, dtype=np.float32)
  # semantic crop with fill 0
  labels = rng.integers(0, 3, size=(h, w))
  value = np.eye(3, dtype=np.float32)[labels].transpose(2, 0, 1).copy()
  proj0 = dmap.MapProjector(**dict(cfg, fill_value=0.))
  tm = dmap.MapBuilder(map_projector=proj0).plot(
    T(depth), value_map=T(value), cam_pose=np.array(cfg["cam_pose"], dtype=np.float32))
  crop = tm.select(torch.tensor([[20, 40]]), 32, 32)
  arrays["sem_value"] = value
  arrays["sem_crop_map"] = crop.topdown_map.numpy().copy()
  arrays["sem_crop_mask"] = crop.mask.numpy().copy()
  arrays["sem_crop_height"] = crop.height_map.contiguous().numpy().copy()
  save("g9_topdownmap_queries", **pack_kwargs(cfg), **arrays)


def g5b(dmap):
  """Eight seeded random MapBuilder.step(merge=True) sequences of four frames (maps.py:2357-2508,
  2181-2287): local and global step maps, height and semantic, the three centre modes, two
  resolutions, keep_pose -- the world map (content, size, offsets) after every step."""
  h, w = 40, 56
  hfov, pitch, camh = np.radians(70.), np.radians(-20.), 0.88
  arrays, meta = {}, []
  for s in range(8):
    rng = np.random.default_rng(5500 + s)
    semantic = s % 2 == 1
    to_global = bool((s >> 1) & 1)
    mode = ("none", "origin", "camera")[s % 3]
    res = (0.05, 0.1)[(s >> 2) & 1]
    keep_pose = s == 5
    mw, mh = (48, 40) if s % 4 == 0 else (64, 64)
    cfg = dict(width=w, height=h, hfov=hfov, cam_pose=[0., 0., 0.], width_offset=mw / 2.,
               height_offset=0. if not to_global else mh / 2., cam_pitch=pitch, cam_height=camh,
               map_res=res, map_width=mw, map_height=mh, trunc_depth_min=0.15,
               trunc_depth_max=5.05, clip_border=s % 3, fill_value=0. if semantic else -np.inf,
               to_global=to_global)
    proj = dmap.MapProjector(**cfg)
    build = dmap.MapBuilder(map_projector=proj)
    build.reset()
    poses = np.stack([rng.uniform(-1.2, 1.2, 4), rng.uniform(-1.2, 1.2, 4),
                      rng.uniform(-np.pi, np.pi, 4)], 1).astype(np.float32)
    poses[0] = 0.
    C = 3
    for t in range(4):
      depth = scene_depth(rng, h, w, hfov, pitch, camh)[None]
      value = None
      if semantic:
        labels = rng.integers(0, C, size=(h, w))
        value = np.eye(C, dtype=np.float32)[labels].transpose(2, 0, 1).copy()
      local = build.step(depth_map=T(depth), value_map=None if value is None else T(value),
                         cam_pose=poses[t].copy(), center_mode=mode, merge=True,
                         keep_pose=keep_pose and t > 0)
      wm = build.world_map
      k = f"s{s}_"
      arrays[k + f"depth{t}"] = depth
      if semantic:
        arrays[k + f"labels{t}"] = labels.astype(np.uint8)
      arrays[k + f"local_map{t}"] = local.topdown_map.numpy().copy()
      arrays[k + f"local_mask{t}"] = np.packbits(local.mask.numpy())
      arrays[k + f"world_map{t}"] = wm.topdown_map.numpy().copy()
      arrays[k + f"world_mask{t}"] = np.packbits(wm.mask.numpy())
      arrays[k + f"world_height{t}"] = wm.height_map.contiguous().numpy()[:, :1].copy()
      arrays[k + f"world_woff{t}"] = np.asarray(wm.proj.width_offset, dtype=np.float32)
      arrays[k + f"world_hoff{t}"] = np.asarray(wm.proj.height_offset, dtype=np.float32)
      arrays[k + f"world_size{t}"] = np.array([wm.proj.map_width, wm.proj.map_height])
      arrays[k + f"world_pose{t}"] = np.asarray(wm.proj.cam_pose, dtype=np.float32).reshape(-1)
    arrays[f"s{s}_poses"] = poses
    meta.append([int(semantic), int(to_global), ("none", "origin", "camera").index(mode),
                 int(round(res * 100)), int(keep_pose), mw, mh, s % 3])
  save("g5b_builder_random_sequences", meta=np.asarray(meta), height=np.int64(h), width=np.int64(w),
       hfov=np.float64(hfov), pitch=np.float64(pitch), cam_height=np.float64(camh), **arrays)


def g9b(dmap):
  """TopdownMap.select on a 2048 x 2048 map (BASELINE configs[4]'s map size; maps.py:1959-2037,
  utils.py:571-652): 1024 x 1024 and odd-sized crops at integral, fractional and out-of-range
  centres.  The map is regenerated from the seed by the test; the fixture keeps every 8th
  row / column of each crop plus its checksum and the shifted offsets."""
  seed, n, stride = 9009, 2048, 8
  rng = np.random.default_rng(seed)
  top = rng.uniform(-1.0, 3.0, (1, 1, n, n)).astype(np.float32)
  mask = rng.uniform(size=(1, 1, n, n)) > 0.35
  top[~mask] = -np.inf
  cfg = dict(width=1280, height=960, hfov=np.radians(70.), cam_pose=[0.4, -0.3, 0.7],
             width_offset=1024., height_offset=1024., cam_pitch=np.radians(-20.),
             cam_height=0.88, map_res=0.03, map_width=n, map_height=n, to_global=True,
             fill_value=-np.inf)
  proj = dmap.MapProjector(**cfg)
  tm = dmap.TopdownMap(topdown_map=T(top), mask=torch.from_numpy(mask), height_map=T(top),
                       map_projector=proj)
  arrays = {}
  cases = [([1024., 1024.], 1024, 1024), ([1037.25, 991.75], 1024, 1024), ([100.5, 2000.49], 1024, 1024),
           ([-300., 2500.], 1024, 1024), ([1500.5, 700.5], 777, 1001), ([2047., 0.], 512, 2048)]
  for i, (center, cw, ch) in enumerate(cases):
    crop = tm.select(T([center]), cw, ch)
    cm, ck = crop.topdown_map.numpy(), crop.mask.numpy()
    arrays[f"c{i}_center"] = np.asarray(center, dtype=np.float32)
    arrays[f"c{i}_size"] = np.array([cw, ch])
    arrays[f"c{i}_map"] = cm[:, :, ::stride, ::stride].copy()
    arrays[f"c{i}_mask"] = ck[:, :, ::stride, ::stride].copy()
    finite = np.isfinite(cm)
    arrays[f"c{i}_sum"] = np.float64(cm[finite].astype(np.float64).sum())
    arrays[f"c{i}_cells"] = np.array([int(finite.sum()), int(ck.sum())])
    arrays[f"c{i}_woff"] = np.asarray(crop.proj.width_offset, dtype=np.float32)
    arrays[f"c{i}_hoff"] = np.asarray(crop.proj.height_offset, dtype=np.float32)
  save("g9b_crop_2048_sampled", **pack_kwargs(cfg), seed=np.int64(seed), stride=np.int64(stride),
       ncases=np.int64(len(cases)), top_checksum=np.float64(top[np.isfinite(top)].astype(np.float64).sum()),
       **arrays)


def g10(dmap):
  """Docstring known answers (utils.py:340-351, 62-67; maps.py:34-39) and
  Rodrigues matrices / intrinsics as the reference computes them."""
  arrays = {}
  arrays["ravel_in"] = np.array([[3, 2, 3], [0, 2, 1]])
  arrays["ravel_out"] = dmap.utils.ravel_index(
    torch.tensor([[3, 2, 3], [0, 2, 1]]), (6, 5, 4)).numpy().copy()
  assert arrays["ravel_out"].tolist() == [71, 9]
  assert dmap.Reduction(None) == dmap.Reduction.max
  assert dmap.CenterMode(None) == dmap.CenterMode.none
  # rotation matrices: rotate the identity basis (utils.py:261-330)
  rng = np.random.default_rng(1010)
  angles = np.concatenate([
    rng.uniform(-np.pi, np.pi, size=200),
    [0.0, 0.001, -0.001, 0.0011, 0.0009, np.pi, -np.pi, np.pi / 2],
    np.radians([-20., -15., -25., -31., -10.])]).astype(np.float32)
  eye = torch.eye(3).view(1, 3, 3)
  Rx, Ry = [], []
  for a in angles:
    Rx.append(dmap.utils.rotate(eye.clone(), [1., 0., 0.], T([a])).numpy()[0])
    Ry.append(dmap.utils.rotate(eye.clone(), [0., 1., 0.], T([a])).numpy()[0])
  # rotate(e_j)[i] = R[j][i]  ->  row j of output is row j of R
  arrays["angles"] = angles
  arrays["Rx"] = np.stack(Rx)
  arrays["Ry"] = np.stack(Ry)
  intr = []
  for (w, h, hf, vf) in ((320, 240, 70., None), (640, 480, 70., None),
                         (64, 48, 79., 55.), (1280, 960, 90., None)):
    ci = dmap.utils.get_camera_intrinsics(
      w, h, np.radians(hf), None if vf is None else np.radians(vf))
    intr.append([w, h, hf, -1 if vf is None else vf, ci.cx, ci.cy, ci.fx, ci.fy])
  arrays["intrinsics"] = np.array(intr, dtype=np.float64)
  save("g10_known_answers", **arrays)


def g11(dmap):
  """Seeded sweep over the projector's parameter space: 12 configurations x 2 frames,
  each frame one B=1 reference call (pitch of either sign, non-square pixels, offsets,
  flips, local/global, border, truncations, valid maps, min/max, finite fills)."""
  rng = np.random.default_rng(1111)
  arrays = {}
  for k in range(12):
    B, h, w = 2, 40, 56
    mh, mw = [(48, 64), (64, 48), (56, 56)][k % 3]
    hfov = float(rng.uniform(0.8, 1.7))
    pitch = rng.uniform(-0.7, 0.4, size=B).astype(np.float32)
    camh = rng.uniform(0.3, 1.5, size=B).astype(np.float32)
    if k % 2:
      depth = np.stack([scene_depth(rng, h, w, hfov, float(pitch[b]), float(camh[b]))
                        for b in range(B)])[:, None]
    else:
      depth = rng.uniform(0.1, 8.0, size=(B, 1, h, w)).astype(np.float32)
    pose = np.stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B),
                     rng.uniform(-np.pi, np.pi, B)], axis=1).astype(np.float32)
    is_max = bool(k % 4 != 3)
    cfg = dict(width=w, height=h, hfov=hfov,
               vfov=None if k % 2 else float(rng.uniform(0.7, 1.4)),
               map_res=float(rng.choice([0.05, 0.08, 0.1])), map_width=mw, map_height=mh,
               trunc_depth_min=float(rng.choice([0.0, 0.15, 0.5])),
               trunc_depth_max=float(rng.choice([2.5, 5.05, 7.0])),
               trunc_height_max=None if k % 3 else float(rng.uniform(0.2, 1.2)),
               clip_border=int(rng.choice([0, 0, 3])),
               to_global=bool(k % 5 != 0), flip_h=bool(k % 6 != 1),
               fill_value=(-np.inf if is_max else np.inf) if k % 3 else float(rng.uniform(-1, 1)),
               reduction="max" if is_max else "min")
    woff = (mw / 2 + rng.uniform(-10, 10, size=B)).astype(np.float32)
    hoff = (mh / 2 + rng.uniform(-10, 10, size=B)).astype(np.float32)
    valid = (rng.uniform(size=(B, 1, h, w)) > 0.1) if k % 4 == 2 else None
    tops, masks = [], []
    for b in range(B):
      call = dict(cam_pose=pose[b], cam_pitch=pitch[b:b + 1], cam_height=camh[b:b + 1],
                  width_offset=woff[b:b + 1], height_offset=hoff[b:b + 1])
      r = run_orth(dmap, cfg, depth[b], valid=None if valid is None else valid[b],
                   call_kwargs=call, intermediates=False)
      tops.append(r["topdown"]); masks.append(r["mask"])
    arrays.update({f"k{k}_depth": depth, f"k{k}_cam_pose": pose, f"k{k}_cam_pitch": pitch,
                   f"k{k}_cam_height": camh, f"k{k}_width_offset": woff,
                   f"k{k}_height_offset": hoff, f"k{k}_topdown": np.concatenate(tops, axis=0),
                   f"k{k}_mask": np.concatenate(masks, axis=0)})
    if valid is not None:
      arrays[f"k{k}_valid_map"] = valid
    for name, v in cfg.items():
      arrays[f"k{k}_cfg_{name}"] = np.array(np.nan) if v is None else np.asarray(v)
  save("g11_parameter_sweep_56x40", **arrays)


def g12(dmap):
  """Public method signatures of the object API (maps.py:1253-1749, 1753-1955, 2289-2550):
  parameter names in order, which are required, and the repr of simple defaults."""
  import inspect
  import json
  out = {}
  for cls in (dmap.MapProjector, dmap.TopdownMap, dmap.MapBuilder):
    for name, fn in inspect.getmembers(cls, predicate=inspect.isfunction):
      if name.startswith("_") and name != "__init__":
        continue
      out[f"{cls.__name__}.{name}"] = [
          {"name": p.name, "required": p.default is inspect.Parameter.empty
                                       and p.kind is not inspect.Parameter.VAR_KEYWORD,
           "kind": p.kind.name,
           "default": None if p.default is inspect.Parameter.empty else repr(p.default)}
          for p in inspect.signature(fn).parameters.values()]
  with open(os.path.join(HERE, "g12_api_signatures.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print("wrote g12_api_signatures.json", len(out), "methods")


def g13(dmap):
  """depth_map_to_point_cloud (maps.py:462-545) and height_map_to_point_cloud (maps.py:547-612) as
  callers use them on their own: the camera-space cloud + validity of a depth map (both flips, a valid
  map, truncation on / off) and the cell centres of a height map (both flips, per-frame offsets)."""
  rng = np.random.default_rng(1313)
  h, w = 24, 32
  depth = rng.uniform(0.05, 7.0, (2, 1, h, w)).astype(np.float32)
  depth[0, 0, 0, :4] = [0.0, np.nan, np.inf, -1.0]
  valid = rng.uniform(size=(2, 1, h, w)) > 0.2
  i = dmap.utils.get_camera_intrinsics(width=w, height=h, hfov=np.radians(70.))
  arrays = dict(depth=depth, valid=valid, intr=np.array([i.cx, i.cy, i.fx, i.fy], dtype=np.float64))
  for tag, kw in (("flip", dict(flip_h=True, trunc_depth_min=0.15, trunc_depth_max=5.05, valid_map=T(valid, torch.bool))),
                  ("noflip", dict(flip_h=False, trunc_depth_min=None, trunc_depth_max=None, valid_map=None))):
    cloud, ok = dmap.maps.depth_map_to_point_cloud(
      depth_map=T(depth), focal_x=i.fx, focal_y=i.fy, center_x=i.cx, center_y=i.cy, **kw)
    arrays[f"cloud_{tag}"] = cloud.numpy().copy()
    arrays[f"ok_{tag}"] = ok.numpy().copy()
  hm = rng.uniform(-1.0, 2.0, (2, 3, 20, 28)).astype(np.float32)
  hm[0, 0, 0, 0] = -np.inf
  woff = np.array([14.0, 3.25], dtype=np.float32)
  hoff = np.array([0.0, 10.5], dtype=np.float32)
  arrays.update(height=hm, woff=woff, hoff=hoff)
  for tag, flip in (("flip", True), ("noflip", False)):
    pc = dmap.maps.height_map_to_point_cloud(height_map=T(hm), width_offset=T(woff), height_offset=T(hoff),
                                             map_res=0.07, map_height=20, flip_h=flip)
    arrays[f"points_{tag}"] = pc.numpy().copy()
  save("g13_point_clouds", **arrays)


def g9c(dmap):
  """crop_topdown_map with the interpolating modes (maps.py:1959-2037; utils.image_sample with
  mode='bilinear' / 'bicubic', utils.py:613-652): a height map with empty (-inf) cells -- interpolation next
  to them gives inf / NaN exactly where the reference's arithmetic does -- and an all-finite value map with
  fill None ('zeros' padding) and a finite fill ('border'); fractional, integral and out-of-range centres."""
  rng = np.random.default_rng(9393)
  n_h, n_w = 40, 56
  top = rng.uniform(-1.0, 3.0, (2, 1, n_h, n_w)).astype(np.float32)
  mask = rng.uniform(size=(2, 1, n_h, n_w)) > 0.2
  height = top.copy()
  height[~mask] = -np.inf
  value = rng.uniform(0.0, 2.0, (2, 3, n_h, n_w)).astype(np.float32)
  cfg = dict(width=64, height=48, hfov=np.radians(70.), cam_pose=[0., 0., 0.], width_offset=28.,
             height_offset=20., cam_pitch=np.radians(-20.), cam_height=0.88, map_res=0.05,
             map_width=n_w, map_height=n_h, to_global=True, fill_value=-np.inf)
  arrays = dict(height=height, mask=mask, value=value)
  centers = np.array([[[27.25, 19.5], [10.0, 30.0]], [[55.75, -0.25], [-3.5, 41.0]], [[28.0, 20.0], [28.5, 20.5]]],
                     dtype=np.float32)
  arrays["centers"] = centers
  for mode in ("bilinear", "bicubic"):
    for ci, center in enumerate(centers):
      tm = dmap.TopdownMap(topdown_map=T(height), mask=torch.from_numpy(mask), height_map=T(height),
                           map_projector=dmap.MapProjector(**cfg))
      crop = dmap.maps.crop_topdown_map(tm, T(center).clone(), 24, 20, mode=mode)
      arrays[f"{mode}_h{ci}_map"] = crop.topdown_map.numpy().copy()
      arrays[f"{mode}_h{ci}_mask"] = crop.mask.numpy().copy()
      for tag, fill in (("none", None), ("half", 0.5)):
        proj = dmap.MapProjector(**dict(cfg, fill_value=fill))
        tv = dmap.TopdownMap(topdown_map=T(value), mask=torch.from_numpy(np.broadcast_to(mask, value.shape).copy()),
                             height_map=T(np.broadcast_to(height, value.shape).copy()), map_projector=proj,
                             is_height_map=False)
        cv = dmap.maps.crop_topdown_map(tv, T(center).clone(), 24, 20, fill_value=fill, mode=mode)
        arrays[f"{mode}_v{ci}_{tag}"] = cv.topdown_map.numpy().copy()
  save("g9c_crop_interpolated", **pack_kwargs(cfg), **arrays)


def main():
  torch.set_num_threads(1)
  torch.manual_seed(0)
  dmap = _import_reference()
  if os.environ.get("DM_GOLDEN_ONLY"):       # e.g. DM_GOLDEN_ONLY=g5b,g9b: only these fixtures
    for name in os.environ["DM_GOLDEN_ONLY"].split(","):
      globals()[name](dmap)
    return
  g1_g2(dmap)
  g3(dmap)
  g4(dmap)
  g5(dmap)
  g6(dmap)
  g7(dmap)
  g8(dmap)
  g8b(dmap)
  g9(dmap)
  g5b(dmap)
  g9b(dmap)
  g10(dmap)
  g11(dmap)
  g12(dmap)
  g13(dmap)
  g9c(dmap)


if __name__ == "__main__":
  main()
