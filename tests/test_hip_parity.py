"""GPU parity: the HIP path (through the public API -> ctypes -> C ABI) against
(a) the reference-generated golden fixtures and (b) the CPU oracle on seeded
synthetic inputs.

Bar (north_star): integer cell indices bit-exact -- checked through the masks
and through exact equality of max/min maps; float heights within 1e-5 (they
are in fact bit-equal); order-dependent sum/mean/prod within rtol 1e-5.
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

from conftest import assert_masks_equal_away_from_fill, load_golden, project_kwargs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dmap():
  import dungeon_maps_amd as dmap
  from dungeon_maps_amd import _native
  _native.lib()  # the HIP library must be the thing under test
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU (run with -m gpu on an MI355X box)")
  return dmap


def _run(dmap, cfg, depth, value=None, valid=None, get_height_map=False, **call):
  proj = dmap.MapProjector(**cfg)
  dev = torch.device("cuda:0")
  outs = proj.orth_project(
      torch.from_numpy(depth).to(dev),
      value_map=None if value is None else torch.from_numpy(value).to(dev),
      valid_map=None if valid is None else torch.from_numpy(valid).to(dev),
      get_height_map=get_height_map, **call)
  torch.cuda.synchronize()
  return [o.cpu().numpy() for o in outs]


def _proj_cfg(cfg):
  c = dict(cfg)
  return c


@pytest.mark.parametrize("name,height", [
    ("g1_height_320x240_256", True),
    ("g2_local_noflip_clip_64x48", False),
    ("g6a_edge_depth_notrunc", True),
    ("g6b_edge_depth_trunc", True),
    ("g6c_fill_none", False),
])
def test_golden_height_maps(dmap, name, height):
  g, cfg = load_golden(name)
  outs = _run(dmap, _proj_cfg(cfg), g["depth"], get_height_map=height)
  np.testing.assert_array_equal(outs[1], g["mask"])
  np.testing.assert_array_equal(outs[0], g["topdown"])
  if height:
    np.testing.assert_array_equal(outs[2], g["height"])


@pytest.mark.parametrize("name", ["g3_semantic_onehot5_64x48", "g3b_semantic_validmap_64x48"])
def test_golden_semantic(dmap, name):
  g, cfg = load_golden(name)
  outs = _run(dmap, _proj_cfg(cfg), g["depth"], value=g["value"], valid=g.get("valid_map"),
              get_height_map=True)
  np.testing.assert_array_equal(outs[1], g["mask"])
  np.testing.assert_array_equal(outs[0], g["topdown"])
  np.testing.assert_array_equal(outs[2], g["height"])


def test_golden_batched(dmap):
  g, cfg = load_golden("g4_batched4_64x48")
  outs = _run(dmap, _proj_cfg(cfg), g["depth"], get_height_map=True,
              cam_pose=g["cam_pose"], cam_pitch=g["cam_pitch"], cam_height=g["cam_height"],
              width_offset=g["width_offset"], height_offset=g["height_offset"])
  np.testing.assert_array_equal(outs[1], g["mask"])
  np.testing.assert_array_equal(outs[0], g["topdown"])
  assert outs[2] is not None


def test_golden_reductions(dmap):
  g, cfg = load_golden("g7_reductions_64x48")
  for red, fill in (("min", np.inf), ("min", 0.0), ("sum", 0.0), ("sum", 1.0),
                    ("mean", 0.0), ("mean", 2.0), ("prod", 1.0), ("max", 0.0),
                    ("max", 1.0)):
    c = dict(cfg, fill_value=fill, reduction=red)
    tag = f"{red}_fill{fill}"
    for prefix, value in (("", g["value"]), ("height_", None)):
      outs = _run(dmap, c, g["depth"], value=value)
      want = g[f"{prefix}topdown_{tag}"]
      if red in ("max", "min"):
        np.testing.assert_array_equal(outs[0], want)
        np.testing.assert_array_equal(outs[1], g[f"{prefix}mask_{tag}"])
      else:
        np.testing.assert_allclose(outs[0], want, rtol=1e-5, atol=1e-6)
        # (positive values cannot cancel: their masks are exact; sums of heights may land on
        # either side of the fill value only where the reference's own sum is within rounding of it)
        assert_masks_equal_away_from_fill(outs[1], g[f"{prefix}mask_{tag}"], want, fill,
                                          exact=(prefix == ""), what=tag)


# --------------------------------------------------------------------------
# seeded synthetic inputs vs the oracle
# --------------------------------------------------------------------------
def _synthetic(B, H, W, seed, scene=False):
  g = torch.Generator().manual_seed(seed)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g)
  pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
  pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
  return depth.numpy(), pose.numpy()


def _oracle_kwargs(oracle, cfg):
  return project_kwargs(cfg, oracle.camera_intrinsics)


@pytest.mark.parametrize("B,H,W,mh,mw", [
    (3, 48, 64, 64, 64),
    (2, 51, 67, 33, 95),       # ragged: nothing divisible by 4
    (1, 240, 320, 256, 256),   # BASELINE configs[0]
    (4, 480, 640, 512, 512),   # BASELINE configs[1] geometry, small batch
])
def test_random_depth_vs_oracle(dmap, oracle, B, H, W, mh, mw):
  depth, pose = _synthetic(B, H, W, seed=1234 + B)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2., map_res=0.03,
             map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05,
             to_global=True, fill_value=-np.inf)
  outs = _run(dmap, cfg, depth, cam_pose=pose)
  want = oracle.orth_project(depth, **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose),
                             nthreads=8)
  np.testing.assert_array_equal(outs[1], want[1])
  np.testing.assert_array_equal(outs[0], want[0])


@pytest.mark.parametrize("k", range(12))
def test_golden_parameter_sweep(dmap, k):
  """g11: 12 seeded configurations run by the reference itself; HIP path bit-equal."""
  from conftest import load_sweep
  g, cfg = load_sweep()[k]
  call = {n: g[n] for n in ("cam_pose", "cam_pitch", "cam_height", "width_offset", "height_offset")}
  outs = _run(dmap, cfg, g["depth"], valid=g.get("valid_map"), **call)
  np.testing.assert_array_equal(outs[1], g["mask"])
  np.testing.assert_array_equal(outs[0], g["topdown"])


@pytest.mark.parametrize("seed", range(24))
def test_random_configurations_vs_oracle(dmap, oracle, seed):
  """Seeded sweep over the projector's parameter space (pitch of either sign, non-square
  pixels, resolutions, offsets, flips, local/global, borders, truncations, valid maps,
  per-frame pitch/height, min/max, scene-like and uniform depth, batch sizes that pick
  different strip layouts): GPU == oracle bit for bit."""
  rng = np.random.default_rng(1000 + seed)
  B = int(rng.choice([1, 2, 5, 17, 40]))
  H, W = [(48, 64), (60, 80), (96, 128), (50, 70)][int(rng.integers(4))]
  mh, mw = [(64, 64), (96, 128), (128, 96), (160, 160)][int(rng.integers(4))]
  scene = bool(rng.integers(2))
  if scene:     # floor + walls: many pixels per cell
    rows = np.arange(H, dtype=np.float64).reshape(1, 1, H, 1)
    wall = rng.uniform(1.0, 6.0, size=(B, 1, 1, W // 8 + 1)).repeat(8, axis=3)[..., :W]
    floor = 0.9 / np.maximum(0.05, (rows - H * 0.45) / (H * 0.9))
    depth = np.minimum(floor, wall).astype(np.float32)
    depth = np.broadcast_to(depth, (B, 1, H, W)).copy()
  else:
    depth = rng.uniform(0.1, 8.0, size=(B, 1, H, W)).astype(np.float32)
  pose = np.stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), rng.uniform(-np.pi, np.pi, B)],
                  axis=1).astype(np.float32)
  per_frame = bool(rng.integers(2))
  pitch = rng.uniform(-0.7, 0.4, size=B if per_frame else 1).astype(np.float32)
  camh = rng.uniform(0.3, 1.5, size=B if per_frame else 1).astype(np.float32)
  is_max = bool(rng.integers(4))           # mostly max
  cfg = dict(width=W, height=H, hfov=float(rng.uniform(0.8, 1.7)),
             vfov=None if rng.integers(2) else float(rng.uniform(0.7, 1.4)),
             cam_pitch=pitch, cam_height=camh,
             width_offset=float(mw / 2 + rng.uniform(-20, 20)),
             height_offset=float(mh / 2 + rng.uniform(-20, 20)),
             map_res=float(rng.choice([0.03, 0.05, 0.08, 0.1])), map_width=mw, map_height=mh,
             trunc_depth_min=float(rng.choice([0.0, 0.15, 0.5])),
             trunc_depth_max=float(rng.choice([2.5, 5.05, 7.0])),
             trunc_height_max=None if rng.integers(3) else float(rng.uniform(0.2, 1.2)),
             clip_border=int(rng.choice([0, 0, 3, 9])),
             to_global=bool(rng.integers(2)), flip_h=bool(rng.integers(4)),
             fill_value=(-np.inf if is_max else np.inf) if rng.integers(3) else float(rng.uniform(-1, 1)),
             reduction="max" if is_max else "min")
  valid = (rng.uniform(size=(B, 1, H, W)) > 0.1) if rng.integers(3) == 0 else None
  outs = _run(dmap, cfg, depth, valid=valid, cam_pose=pose)
  want = oracle.orth_project(depth, valid_map=valid,
                             **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose), nthreads=8)
  np.testing.assert_array_equal(outs[1], want[1])
  np.testing.assert_array_equal(outs[0], want[0])


def test_semantic_40_classes_vs_oracle(dmap, oracle):
  """BASELINE configs[2] geometry at a small batch: 40-class one-hot, fill 0."""
  B, H, W, C = 2, 120, 160, 40
  depth, pose = _synthetic(B, H, W, seed=99)
  g = torch.Generator().manual_seed(5)
  labels = torch.randint(0, C, (B, H, W), generator=g)
  value = torch.nn.functional.one_hot(labels, C).permute(0, 3, 1, 2).float().contiguous().numpy()
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=64., height_offset=64., map_res=0.06,
             map_width=128, map_height=128, trunc_depth_min=0.15, trunc_depth_max=5.05,
             to_global=True, fill_value=0.0)
  outs = _run(dmap, cfg, depth, value=value, get_height_map=True, cam_pose=pose)
  want = oracle.orth_project(depth, value_map=value, get_height_map=True,
                             **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose))
  np.testing.assert_array_equal(outs[1], want[1])
  np.testing.assert_array_equal(outs[0], want[0])
  np.testing.assert_array_equal(outs[2], np.ascontiguousarray(want[2]))


def test_empty_and_degenerate_inputs(dmap, oracle):
  cfg = dict(width=8, height=4, hfov=np.radians(70.), cam_pitch=0.0, cam_height=1.0,
             width_offset=4., height_offset=4., map_res=0.5, map_width=8, map_height=8,
             to_global=False, fill_value=-np.inf)
  # all pixels invalid: map stays at fill, mask all False
  depth = np.full((1, 1, 4, 8), np.nan, dtype=np.float32)
  top, mask = _run(dmap, cfg, depth)
  assert not mask.any() and np.isneginf(top).all()
  # 2-D and 3-D inputs are promoted like the reference's to_4D_image
  d2 = np.full((4, 8), 2.0, dtype=np.float32)
  top2, mask2 = _run(dmap, cfg, d2)
  assert top2.shape == (1, 1, 8, 8)
  want = oracle.orth_project(d2, **_oracle_kwargs(oracle, cfg))
  np.testing.assert_array_equal(top2, want[0])
  np.testing.assert_array_equal(mask2, want[1])


def test_cpu_tensors_round_trip_through_gpu(dmap, oracle):
  depth, pose = _synthetic(1, 48, 64, seed=3)
  cfg = dict(width=64, height=48, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=32., height_offset=32., map_res=0.1,
             map_width=64, map_height=64, to_global=True, fill_value=-np.inf)
  proj = dmap.MapProjector(**cfg)
  top, mask = proj.orth_project(depth, cam_pose=pose)     # numpy in
  assert top.device.type == "cpu" and mask.dtype == torch.bool
  want = oracle.orth_project(depth, **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose))
  np.testing.assert_array_equal(top.numpy(), want[0])


def test_mean_of_one_hot_value_maps_on_the_window_path(dmap, oracle):
  """reduction='mean' of a 3-class one-hot value map, 9 frames, border clipped, no far truncation
  (the window path's kernel variant that runs short of registers): equal to the oracle, cell
  for cell, three times in a row.  This is seed 1401484 of the parity campaign's mean mode, which
  came out with the image-row number in the first cell of some float4 groups of the FILL region:
  a buffer_store_dwordx4 with its scalar offset in a register, followed at once by an instruction
  that overwrites its data registers (dm_pixel.hpp buffer_store_b128_at_scalar_offset)."""
  rng = np.random.default_rng(50_000 + 1401484)
  B, H, W, mh, mw, C = 9, 48, 64, 96, 128, 3
  depth = rng.uniform(0.1, 8.0, size=(B, 1, H, W)).astype(np.float32)
  pose = np.stack([rng.uniform(-2, 2, B), rng.uniform(-2, 2, B), rng.uniform(-np.pi, np.pi, B)],
                  axis=1).astype(np.float32)
  value = np.eye(C, dtype=np.float32)[rng.integers(0, C, size=(B, H, W))].transpose(0, 3, 1, 2).copy()
  cfg = dict(width=W, height=H, hfov=1.7576597295876581, vfov=None,
             cam_pitch=rng.uniform(-0.9, 0.5, size=B).astype(np.float32),
             cam_height=rng.uniform(0.2, 2.0, size=B).astype(np.float32),
             width_offset=29.19671036767047, height_offset=79.95770080230739, map_res=0.0625,
             map_width=mw, map_height=mh, trunc_depth_min=0.5, trunc_depth_max=None,
             trunc_height_max=1.0342564776427254, clip_border=9, to_global=False, flip_h=True,
             fill_value=1.0, reduction="mean")
  want = oracle.orth_project(depth, value_map=value, nthreads=8,
                             **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose))
  for _ in range(3):
    # (what the allocator hands out next holds other numbers each time)
    junk = torch.full((B * C * mh * mw,), 12345.0, device="cuda"); del junk
    outs = _run(dmap, cfg, depth, value=value, cam_pose=pose)
    np.testing.assert_array_equal(outs[1], want[1])
    np.testing.assert_array_equal(outs[0], want[0])


@pytest.mark.parametrize("B", [65, 129])
def test_fused_projection_of_more_frames_than_the_slab_workspace_holds(dmap, oracle, B):
  """dm_orth_project_fused_f32 on the window path (70 pixel columns: no column strips) with one
  frame more than the workspace holds slabs for: the frames go through in halves, the second
  half folding into the map of the first.  (B = 65 used to fail with "HIP launch failed: out
  of memory": found by tests/campaigns/parity_campaign.py in its fused mode.)"""
  rng = np.random.default_rng(B)
  H, W = 50, 70
  depth = rng.uniform(0.1, 8.0, size=(B, 1, H, W)).astype(np.float32)
  pose = np.stack([rng.uniform(-2, 2, B), rng.uniform(-2, 2, B), rng.uniform(-3, 3, B)], axis=1).astype(np.float32)
  cfg = dict(width=W, height=H, hfov=0.947, cam_pitch=-0.3, cam_height=0.9, width_offset=128.,
             height_offset=128., map_res=0.02, map_width=256, map_height=256, trunc_depth_min=0.5,
             trunc_depth_max=5.05, to_global=True, flip_h=False, fill_value=-np.inf)
  proj = dmap.MapProjector(**cfg)
  fused, fmask = proj.orth_project_fused(torch.from_numpy(depth).cuda(), cam_pose=pose)
  torch.cuda.synchronize()
  want = oracle.orth_project(depth, fused=True, nthreads=8,
                             **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose))
  np.testing.assert_array_equal(fmask.cpu().numpy(), want[1])
  np.testing.assert_array_equal(fused.cpu().numpy(), want[0])
  # ... and into a running map (accumulate) with min / a finite fill value
  cfg2 = dict(cfg, fill_value=3.0, reduction="min")
  proj2 = dmap.MapProjector(**cfg2)
  acc, _ = proj2.orth_project_fused(torch.from_numpy(depth[:5]).cuda(), cam_pose=pose[:5])
  acc, amask = proj2.orth_project_fused(torch.from_numpy(depth[5:]).cuda(), cam_pose=pose[5:], out=acc)
  want2 = oracle.orth_project(depth, fused=True, nthreads=8,
                              **dict(_oracle_kwargs(oracle, cfg2), cam_pose=pose))
  np.testing.assert_array_equal(amask.cpu().numpy(), want2[1])
  np.testing.assert_array_equal(acc.cpu().numpy(), want2[0])


def test_fused_equals_max_over_frames(dmap, oracle):
  B, H, W = 6, 96, 128
  depth, pose = _synthetic(B, H, W, seed=77)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=128., height_offset=128., map_res=0.05,
             map_width=256, map_height=256, trunc_depth_min=0.15, trunc_depth_max=5.05,
             to_global=True, fill_value=-np.inf)
  proj = dmap.MapProjector(**cfg)
  d = torch.from_numpy(depth).cuda()
  fused, fmask = proj.orth_project_fused(d, cam_pose=pose)
  per_frame, pmask = proj.orth_project(d, cam_pose=pose)
  torch.cuda.synchronize()
  assert torch.equal(fused, per_frame.amax(dim=0))
  assert torch.equal(fmask, pmask.any(dim=0))
  want = oracle.orth_project(depth, fused=True,
                             **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose))
  np.testing.assert_array_equal(fused.cpu().numpy(), want[0])
  np.testing.assert_array_equal(fmask.cpu().numpy(), want[1])
  # the direct fused projection: LDS fast path == generic path
  from dungeon_maps_amd import _native as _nat
  _nat.lib().dm_debug_force_generic_path(1)
  try:
    gfused, gmask = proj.orth_project_fused(d, cam_pose=pose)
  finally:
    _nat.lib().dm_debug_force_generic_path(0)
  assert torch.equal(gfused, fused) and torch.equal(gmask, fmask)
  # running world map: fusing two halves one after the other == fusing all
  acc, _ = proj.orth_project_fused(d[:3], cam_pose=pose[:3])
  acc, amask = proj.orth_project_fused(d[3:], cam_pose=pose[3:], out=acc)
  assert torch.equal(acc, fused) and torch.equal(amask, fmask)
  assert torch.equal(dmap.mask_from_map(fused, -np.inf), fmask)
  # fuse of finished per-frame maps (dm_fuse_batch_f32), incl. odd sizes / running map
  assert torch.equal(dmap.fuse_batch(per_frame, "max"), fused)
  assert torch.equal(dmap.fuse_batch(per_frame[3:], "max", out=dmap.fuse_batch(per_frame[:3])),
                     fused)
  # per-frame maps + fused map from one launch sequence (fast path and generic path)
  from dungeon_maps_amd import _native
  for force in (0, 1):
    _native.lib().dm_debug_force_generic_path(force)
    try:
      t2, m2, f2, fm2 = proj.orth_project_and_fuse(d, cam_pose=pose)
    finally:
      _native.lib().dm_debug_force_generic_path(0)
    assert torch.equal(t2, per_frame) and torch.equal(m2, pmask)
    assert torch.equal(f2, fused) and torch.equal(fm2, fmask)
  odd = torch.randn(5, 3, 7, 11, device="cuda")
  assert torch.equal(dmap.fuse_batch(odd, "min"), odd.amin(dim=0))


def test_full_size_properties(dmap):
  """BASELINE configs[1] at full size (B=64, 640x480 -> 512x512): properties
  that do not need the oracle -- permutation invariance of the frame order,
  B-stack == loop of B=1, mask == f(map, fill), determinism."""
  B, H, W = 64, 480, 640
  depth, pose = _synthetic(B, H, W, seed=1234)
  proj = dmap.MapProjector(
      width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
      width_offset=256., height_offset=256., map_res=0.03, map_width=512, map_height=512,
      trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)
  d = torch.from_numpy(depth).cuda()
  p = torch.from_numpy(pose)
  top, mask = proj.orth_project(d, cam_pose=p)
  top2, mask2 = proj.orth_project(d, cam_pose=p)
  assert torch.equal(top, top2) and torch.equal(mask, mask2)          # deterministic
  perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
  topp, maskp = proj.orth_project(d[perm.cuda()], cam_pose=p[perm])
  assert torch.equal(topp, top[perm.cuda()]) and torch.equal(maskp, mask[perm.cuda()])
  for b in (0, 17, 63):
    t1, m1 = proj.orth_project(d[b:b + 1], cam_pose=p[b:b + 1])
    assert torch.equal(t1[0], top[b]) and torch.equal(m1[0], mask[b])
  assert torch.equal(mask, torch.isfinite(top))                        # F9
  assert mask.any(dim=-1).any(dim=-1).all()                            # every frame hit
  t2, m2, fused, fmask = proj.orth_project_and_fuse(d, cam_pose=p)
  assert torch.equal(t2, top) and torch.equal(m2, mask)
  assert torch.equal(fused, top.amax(dim=0)) and torch.equal(fmask, mask.any(dim=0))
  # more than 64 frames: the fuse walks the union table in chunks of 64
  d3 = torch.cat([d, d[:13]]); p3 = torch.cat([p, p[:13]])
  t3, _, f3, _ = proj.orth_project_and_fuse(d3, cam_pose=p3)
  assert torch.equal(f3, t3.amax(dim=0))


@pytest.mark.parametrize("reduction,fill", [("max", -np.inf), ("min", np.inf), ("max", 0.25)])
def test_batch_fuse_over_many_frames_and_channels(dmap, reduction, fill):
  """The batch fuse (k_fuse_unions) beyond one stage of its union table (more than 256 frames) and over
  several channels: fused == max / min over the batch axis of the per-frame maps, mask == f(map, fill)."""
  B, C, H, W, M = 300, 3, 24, 32, 64
  depth, pose = _synthetic(B, H, W, seed=77)
  g = np.random.default_rng(5)
  value = g.normal(size=(B, C, H, W)).astype(np.float32)
  proj = dmap.MapProjector(
      width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
      width_offset=M / 2., height_offset=M / 2., map_res=0.12, map_width=M, map_height=M,
      trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=fill, reduction=reduction)
  for vm in (None, torch.from_numpy(value).cuda()):
    top, mask, fused, fmask = proj.orth_project_and_fuse(torch.from_numpy(depth).cuda(), value_map=vm,
                                                         cam_pose=torch.from_numpy(pose))
    want = top.amax(dim=0) if reduction == "max" else top.amin(dim=0)
    assert torch.equal(fused, want)
    assert torch.equal(fmask, dmap.mask_from_map(fused, fill))
    assert mask.any()


# --------------------------------------------------------------------------
# LDS-windowed fast path vs the generic global-atomic path (same device)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("case", [
    dict(B=5, H=96, W=128, mh=128, mw=128),                       # strips x bands
    dict(B=3, H=50, W=66, mh=64, mw=64),                          # W % 4 != 0 -> scalar loads
    dict(B=2, H=96, W=128, mh=128, mw=128, flip_h=False, to_global=False),
    dict(B=2, H=96, W=128, mh=128, mw=128, clip_border=7, trunc_height_max=0.6),
    dict(B=2, H=96, W=128, mh=128, mw=128, reduction="min", fill_value=np.inf),
    dict(B=2, H=96, W=128, mh=128, mw=128, fill_value=0.25),      # finite fill takes part
    dict(B=2, H=96, W=128, mh=96, mw=160, trunc_depth_max=None),  # unbounded: window = map
    dict(B=1, H=96, W=128, mh=512, mw=512, trunc_depth_min=None), # window > LDS -> fallback
    dict(B=2, H=96, W=128, mh=128, mw=128, valid=True),
    dict(B=300, H=24, W=32, mh=64, mw=64),                        # more frames than CUs
    dict(B=2, H=96, W=128, mh=128, mw=128, res=1.0 / 3),
    dict(B=2, H=96, W=128, mh=128, mw=128, woff=1000.0),          # frustum misses the map
    dict(B=3, H=96, W=128, mh=128, mw=128, C=5, fill_value=0.0),   # value map, shared index
    dict(B=2, H=96, W=128, mh=128, mw=128, C=3, dc=3, fill_value=-1.0, reduction="min"),
    dict(B=70, H=48, W=64, mh=256, mw=256, C=9, res=0.02),         # many channels: 1 strip too big -> split
    dict(B=2, H=96, W=128, mh=1024, mw=1024, res=0.008),           # long thin wedges, few pixels: generic
    dict(B=3, H=960, W=1280, mh=2048, mw=2048, res=0.01, bands=True),   # long thin wedges: depth bands
    dict(B=2, H=960, W=1280, mh=2048, mw=2048, res=0.01, C=2, fill_value=0.0, bands=True, valid=True),
    dict(B=2, H=960, W=1280, mh=2048, mw=2048, res=0.01, reduction="min", fill_value=np.inf, bands=True),
    dict(B=16, H=960, W=1280, mh=2048, mw=2048, res=0.01, bands=True, halves=True),  # slabs > workspace
    # small images: the cost model would take the generic path, the test hook insists on bands
    dict(B=3, H=120, W=160, mh=512, mw=512, res=0.01, bands=True, force_bands=True),
    dict(B=5, H=96, W=128, mh=512, mw=640, res=0.0125, bands=True, force_bands=True, valid=True,
         clip_border=3, trunc_height_max=0.9, flip_h=False),
    dict(B=2, H=120, W=160, mh=512, mw=512, res=0.0125, C=3, fill_value=0.0, bands=True, force_bands=True),
])
def test_window_path_equals_generic_path(dmap, oracle, case):
  from dungeon_maps_amd import _native
  c = dict(case)
  B, H, W, mh, mw = (c.pop(k) for k in ("B", "H", "W", "mh", "mw"))
  C, dcn = c.pop("C", 0), c.pop("dc", 1)
  use_valid = c.pop("valid", False)
  want_bands, want_halves = c.pop("bands", False), c.pop("halves", False)
  force_bands = c.pop("force_bands", False)
  res = c.pop("res", 0.05)
  woff = c.pop("woff", mw / 2.)
  depth, pose = _synthetic(B, H, W, seed=4321)
  depth[:, :, 0, :3] = [np.nan, np.inf, -1.0]
  if dcn > 1:
    depth = np.concatenate([depth * (1 + 0.1 * k) for k in range(dcn)], axis=1)
  value = None
  if C:
    value = np.random.default_rng(2).normal(size=(B, C, H, W)).astype(np.float32)
    value[:, :, 1, :2] = np.nan                  # NaN values never replace a number
  valid = None
  if use_valid:
    valid = (np.random.default_rng(1).uniform(size=(B, 1, H, W)) > 0.4)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=woff, height_offset=mh / 2., map_res=res,
             map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05,
             to_global=True, fill_value=-np.inf)
  cfg.update(c)
  lib = _native.lib()
  gh = bool(C)
  lib.dm_debug_force_bands(1 if force_bands else 0)
  try:
    fast = _run(dmap, cfg, depth, value=value, valid=valid, get_height_map=gh, cam_pose=pose)
  finally:
    lib.dm_debug_force_bands(0)
  split = (ctypes.c_int32 * 4)()
  lib.dm_debug_last_split(split)
  if want_bands:
    assert split[2] > 1, list(split)
  if want_halves:
    assert split[3] >= 1, list(split)
  lib.dm_debug_force_generic_path(1)
  try:
    slow = _run(dmap, cfg, depth, value=value, valid=valid, get_height_map=gh, cam_pose=pose)
  finally:
    lib.dm_debug_force_generic_path(0)
  for f, s_ in zip(fast, slow):
    np.testing.assert_array_equal(f, s_)
  if B <= 8:
    want = oracle.orth_project(depth, value_map=value, valid_map=valid, get_height_map=gh,
                               **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose))
    np.testing.assert_array_equal(fast[1], want[1])
    np.testing.assert_array_equal(fast[0], want[0])
    if gh:
      np.testing.assert_array_equal(fast[2], np.ascontiguousarray(want[2]))


def test_cfg5_geometry_and_large_batches(dmap, oracle):
  """BASELINE configs[4] geometry (1280x960 depth -> 2048x2048 grid) at a small
  batch vs the oracle, the ego-motion grid at that size, and a batch far larger
  than the CU count."""
  B, H, W, m = 2, 960, 1280, 2048
  depth, pose = _synthetic(B, H, W, seed=55)
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=m / 2., height_offset=m / 2., map_res=0.03,
             map_width=m, map_height=m, trunc_depth_min=0.15, trunc_depth_max=5.05,
             to_global=True, fill_value=-np.inf)
  outs = _run(dmap, cfg, depth, cam_pose=pose)
  want = oracle.orth_project(depth, **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose),
                             nthreads=8)
  np.testing.assert_array_equal(outs[1], want[1])
  np.testing.assert_array_equal(outs[0], want[0])
  proj = dmap.MapProjector(**cfg)
  grid = proj.camera_affine_grid(torch.from_numpy(depth).cuda(), [0.05, 0.1, 0.02])
  assert grid.shape == (B, 1, H, W, 2) and torch.isfinite(grid).all()
  # 1500 small frames: more workgroups than a grid dimension of parts would give
  Bn = 1500
  d, p = _synthetic(Bn, 24, 32, seed=56)
  cfg2 = dict(width=32, height=24, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
              cam_height=0.88, width_offset=32., height_offset=32., map_res=0.2,
              map_width=64, map_height=64, trunc_depth_min=0.15, trunc_depth_max=5.05,
              to_global=True, fill_value=-np.inf)
  top, mask, fused, fmask = dmap.MapProjector(**cfg2).orth_project_and_fuse(
      torch.from_numpy(d).cuda(), cam_pose=p)
  want = oracle.orth_project(d, **dict(_oracle_kwargs(oracle, cfg2), cam_pose=p), nthreads=8)
  np.testing.assert_array_equal(top.cpu().numpy(), want[0])
  np.testing.assert_array_equal(mask.cpu().numpy(), want[1])
  assert torch.equal(fused, top.amax(dim=0)) and torch.equal(fmask, mask.any(dim=0))


# --------------------------------------------------------------------------
# Back-to-back calls with the host running ahead of the GPU: every call stages its own frame
# tables (stream-ordered), so calls in flight must not see each other's poses.
# --------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,mh,mw,calls", [
    (3, 96, 128, 128, 128, 200),       # small calls: dozens of them in flight
    (64, 480, 640, 512, 512, 300),     # cfg2
])
def test_calls_in_flight_keep_their_poses(dmap, B, H, W, mh, mw, calls):
  from dungeon_maps_amd import _native
  lib = _native.lib()
  depth, _ = _synthetic(B, H, W, seed=99)
  depth_d = torch.from_numpy(depth).cuda()
  proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
                           cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2., map_res=0.03,
                           map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05,
                           to_global=True, fill_value=-np.inf)
  g = torch.Generator().manual_seed(7)
  poses = []
  for _ in range(5):                   # five pose sets, visited in turn: every slot sees them all
    pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
    pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
    poses.append(pose)
  lib.dm_debug_force_generic_path(1)
  try:
    refs = [proj.orth_project_and_fuse(depth_d, cam_pose=pose) for pose in poses]
  finally:
    lib.dm_debug_force_generic_path(0)
  bad = torch.zeros((), dtype=torch.int64, device="cuda")
  for i in range(calls):               # no host sync inside the loop
    got = proj.orth_project_and_fuse(depth_d, cam_pose=poses[i % 5])
    ref = refs[i % 5]
    bad += (got[0] != ref[0]).sum() + (got[1] != ref[1]).sum() + (got[2] != ref[2]).sum()
  assert int(bad.item()) == 0
  # the same through the batch-fused entry point
  ref_f = [r[2] for r in refs]
  bad.zero_()
  for i in range(calls // 2):
    fused, _ = proj.orth_project_fused(depth_d, cam_pose=poses[i % 5])
    bad += (fused != ref_f[i % 5]).sum()
  assert int(bad.item()) == 0


def test_two_threads_two_streams(dmap):
  """The library keeps per-thread state (error string, remembered splits): two
  Python threads, each on its own stream and with its own poses, must not disturb each other."""
  import threading
  from dungeon_maps_amd import _native
  lib = _native.lib()
  B, H, W, mh, mw = 6, 96, 128, 128, 128
  depth, _ = _synthetic(B, H, W, seed=11)
  depth_d = torch.from_numpy(depth).cuda()
  proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
                           cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2., map_res=0.03,
                           map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05,
                           to_global=True, fill_value=-np.inf)
  g = torch.Generator().manual_seed(3)
  poses, refs = [], []
  lib.dm_debug_force_generic_path(1)
  try:
    for _ in range(2):
      pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
      pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
      poses.append(pose)
      refs.append(proj.orth_project(depth_d, cam_pose=pose))
  finally:
    lib.dm_debug_force_generic_path(0)
  torch.cuda.synchronize()
  errors = []

  def worker(k):
    try:
      stream = torch.cuda.Stream()
      bad = torch.zeros((), dtype=torch.int64, device="cuda")
      with torch.cuda.stream(stream):
        for _ in range(150):
          top, mask = proj.orth_project(depth_d, cam_pose=poses[k])
          bad += (top != refs[k][0]).sum() + (mask != refs[k][1]).sum()
      stream.synchronize()
      if int(bad.item()) != 0:
        errors.append(f"thread {k}: {int(bad.item())} cells differ")
    except Exception as exc:      # noqa: BLE001 - reported by the main thread
      errors.append(f"thread {k}: {exc!r}")

  threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  assert not errors, errors


def test_two_threads_one_stream(dmap):
  """Two host threads that project on the SAME stream enqueue their launch sequences interleaved (the native
  call runs without the GIL): each call needs a workspace of its own -- the small workspaces the host side keeps
  are kept per thread."""
  import threading
  from dungeon_maps_amd import _native
  lib = _native.lib()
  B, H, W, mh, mw = 2, 96, 128, 128, 128
  depth, _ = _synthetic(B, H, W, seed=12)
  depth_d = torch.from_numpy(depth).cuda()
  proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
                           cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2., map_res=0.03,
                           map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05,
                           to_global=True, fill_value=-np.inf)
  g = torch.Generator().manual_seed(4)
  poses, refs = [], []
  lib.dm_debug_force_generic_path(1)
  try:
    for _ in range(3):
      pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
      pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
      poses.append(pose)
      refs.append(proj.orth_project(depth_d, cam_pose=pose))
  finally:
    lib.dm_debug_force_generic_path(0)
  torch.cuda.synchronize()
  stream = torch.cuda.Stream()
  errors = []

  def worker(k):
    try:
      bad = torch.zeros((), dtype=torch.int64, device="cuda")
      with torch.cuda.stream(stream):
        for _ in range(300):
          top, mask = proj.orth_project(depth_d, cam_pose=poses[k])
          bad += (top != refs[k][0]).sum() + (mask != refs[k][1]).sum()
      stream.synchronize()
      if int(bad.item()) != 0:
        errors.append(f"thread {k}: {int(bad.item())} cells differ")
    except Exception as exc:      # noqa: BLE001 - reported by the main thread
      errors.append(f"thread {k}: {exc!r}")

  threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  assert not errors, errors


@pytest.mark.parametrize("red,fill", [("sum", 0.0), ("mean", 2.0), ("prod", 1.0), ("min", np.inf),
                                      ("max", 0.25)])
def test_generic_path_on_large_sparse_maps(dmap, oracle, red, fill):
  """Large maps at fine resolution: each frame reaches a small part of its map, the generic path
  then finalises (mask, mean division) only inside the frame's union window and the rest of
  the map and mask comes from the fill kernel.  Every reduction, value maps with their height
  map, valid maps, a frame that misses the map."""
  B, H, W, mh, mw = 3, 60, 80, 640, 512
  depth, pose = _synthetic(B, H, W, seed=77)
  pose[2, :2] = [40.0, -35.0]                      # this frame's frustum lies outside the map
  value = np.random.default_rng(5).uniform(0.5, 1.5, size=(B, 2, H, W)).astype(np.float32)
  valid = np.random.default_rng(6).uniform(size=(B, 1, H, W)) > 0.2
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2., map_res=0.01,
             map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=2.05,
             to_global=True, fill_value=fill, reduction=red, clip_border=2)
  got = _run(dmap, cfg, depth, value=value, valid=valid, get_height_map=True, cam_pose=pose)
  want = oracle.orth_project(depth, value_map=value, valid_map=valid, get_height_map=True,
                             **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose))
  if red in ("max", "min"):
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
  else:
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-6)
    assert_masks_equal_away_from_fill(got[1], want[1], want[0], fill, exact=True, what=red)   # (values in [0.5, 1.5]: no cancellation)
  np.testing.assert_array_equal(got[2], np.ascontiguousarray(want[2]))
  assert got[1][2].sum() == 0 and want[1][2].sum() == 0       # the frame that misses the map


@pytest.mark.parametrize("case", [
    dict(B=3, H=96, W=128, mh=97, mw=131),
    dict(B=2, H=96, W=128, mh=128, mw=241, C=2, fill_value=0.0),          # value maps + height map
    dict(B=2, H=50, W=70, mh=63, mw=65, reduction="min", fill_value=np.inf, valid=True),
    dict(B=5, H=96, W=128, mh=101, mw=99, fuse=True),
])
def test_odd_map_widths_stay_on_the_window_path(dmap, oracle, case):
  """mw % 4 != 0 (odd ego-centric maps): projected into maps padded to a multiple of 4 and
  copied out, not sent down the generic path."""
  from dungeon_maps_amd import _native
  lib = _native.lib()
  c = dict(case)
  B, H, W, mh, mw = (c.pop(k) for k in ("B", "H", "W", "mh", "mw"))
  C, use_valid, fuse = c.pop("C", 0), c.pop("valid", False), c.pop("fuse", False)
  depth, pose = _synthetic(B, H, W, seed=808)
  value = np.random.default_rng(2).normal(size=(B, C, H, W)).astype(np.float32) if C else None
  valid = (np.random.default_rng(1).uniform(size=(B, 1, H, W)) > 0.3) if use_valid else None
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2., map_res=0.05,
             map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05,
             to_global=True, fill_value=-np.inf)
  cfg.update(c)
  gh = bool(C)
  got = _run(dmap, cfg, depth, value=value, valid=valid, get_height_map=gh, cam_pose=pose)
  split = (ctypes.c_int32 * 4)()
  lib.dm_debug_last_split(split)
  assert split[0] > 0, "took the generic path"
  want = oracle.orth_project(depth, value_map=value, valid_map=valid, get_height_map=gh,
                             **dict(_oracle_kwargs(oracle, cfg), cam_pose=pose))
  np.testing.assert_array_equal(got[1], want[1])
  np.testing.assert_array_equal(got[0], want[0])
  if gh:
    np.testing.assert_array_equal(got[2], np.ascontiguousarray(want[2]))
  if fuse:
    proj = dmap.MapProjector(**cfg)
    t, m, f, fm = proj.orth_project_and_fuse(torch.from_numpy(depth).cuda(), cam_pose=pose)
    np.testing.assert_array_equal(t.cpu().numpy(), want[0])
    np.testing.assert_array_equal(f.cpu().numpy(), want[0].max(axis=0))
    np.testing.assert_array_equal(fm.cpu().numpy(), want[1].any(axis=0))


@pytest.mark.parametrize("case", [
    dict(B=3, H=96, W=128, mh=128, mw=128),
    dict(B=2, H=50, W=66, mh=64, mw=64),                               # scalar loads, ragged rows
    dict(B=2, H=100, W=160, mh=128, mw=128),                           # strips whose width does not divide 1024
    dict(B=5, H=37, W=96, mh=96, mw=160, clip_border=3, valid=True),   # tail rows, border, valid map
    dict(B=2, H=120, W=160, mh=512, mw=512, res=0.0125, force_bands=True),   # depth bands: shared bounds
    dict(B=2, H=96, W=128, mh=97, mw=131),                             # odd map width: padded route
    dict(B=70, H=48, W=64, mh=128, mw=128),                            # two chunks of frames
])
@pytest.mark.parametrize("red", ["sum", "mean"])
def test_sum_reduction_on_the_window_path(dmap, oracle, case, red):
  """reduction='sum' / 'mean' in LDS windows (mean: a count window beside the sum window, the
  merge divides (fill + sum) by max(count, 1)).  reduction='sum' in LDS windows (ds_add_f32, windows start at 0, fill added once by the merge).
  A sum counts every pixel exactly once -- unlike max / min nothing may be projected twice
  (tail rows, idle threads, depth-band boundaries).  One-hot values make the sums small integers,
  exact in float32 whatever the order: those must be bit-equal to the oracle."""
  from dungeon_maps_amd import _native
  lib = _native.lib()
  c = dict(case)
  B, H, W, mh, mw = (c.pop(k) for k in ("B", "H", "W", "mh", "mw"))
  use_valid, force_bands = c.pop("valid", False), c.pop("force_bands", False)
  res = c.pop("res", 0.05)
  rng = np.random.default_rng(31)
  depth, pose = _synthetic(B, H, W, seed=4242)
  depth.reshape(-1)[rng.integers(0, depth.size, 200)] = rng.choice(
      np.array([0.15, 5.05, 1.375, 2.6, 3.825], np.float32), 200)       # on the (band) bounds
  labels = rng.integers(0, 3, size=(B, H, W))
  onehot = np.eye(3, dtype=np.float32)[labels].transpose(0, 3, 1, 2).copy()
  valid = (rng.uniform(size=(B, 1, H, W)) > 0.3) if use_valid else None
  cfg = dict(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
             cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2., map_res=res,
             map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05,
             to_global=True, fill_value=0.0, reduction=red)
  cfg.update(c)
  lib.dm_debug_force_bands(1 if force_bands else 0)
  try:
    counts = _run(dmap, cfg, depth, value=onehot, valid=valid, get_height_map=True, cam_pose=pose)
    split = (ctypes.c_int32 * 4)()
    lib.dm_debug_last_split(split)
    heights = _run(dmap, dict(cfg, fill_value=1.5), depth, valid=valid, cam_pose=pose)
  finally:
    lib.dm_debug_force_bands(0)
  assert split[0] > 0, "took the generic path"
  if force_bands:
    assert split[2] > 1, list(split)
  kw = dict(_oracle_kwargs(oracle, cfg), cam_pose=pose)
  want = oracle.orth_project(depth, value_map=onehot, valid_map=valid, get_height_map=True, **kw)
  np.testing.assert_array_equal(counts[0], want[0])            # point counts per class: exact
  np.testing.assert_array_equal(counts[1], want[1])
  np.testing.assert_array_equal(counts[2], np.ascontiguousarray(want[2]))
  kw["fill_value"] = 1.5
  want_h = oracle.orth_project(depth, valid_map=valid, **kw)   # sums of heights on top of a fill
  np.testing.assert_allclose(heights[0], want_h[0], rtol=1e-5, atol=1e-5)
  assert_masks_equal_away_from_fill(heights[1], want_h[1], want_h[0], 1.5, what="heights")


def test_three_host_threads_on_their_own_streams(dmap):
  """Three host threads, each on a stream of its own, make different projection calls at the same
  time (64 frames of 240x320, a 3-class and a 5-class value map with its height map, a single
  frame; plain calls and orth_project_and_fuse): every result equals the one the same call gave
  alone.  (ctypes releases the GIL inside the native call: the library's per-thread plan and
  bound caches, the shared status word and the per-call kernel arguments are all in play.)"""
  import threading
  dev = torch.device("cuda:0")

  def make(seed, B, H, W, mh, mw, C):
    rng = np.random.default_rng(seed)
    depth = torch.from_numpy(rng.uniform(0.1, 8.0, size=(B, 1, H, W)).astype(np.float32)).to(dev)
    pose = torch.from_numpy(np.stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B),
                                      rng.uniform(-3, 3, B)], axis=1).astype(np.float32))
    value = None
    if C:
      value = torch.from_numpy(np.eye(C, dtype=np.float32)[rng.integers(0, C, size=(B, H, W))]
                               .transpose(0, 3, 1, 2).copy()).to(dev)
    proj = dmap.MapProjector(width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
                             cam_height=0.88, width_offset=mw / 2., height_offset=mh / 2., map_res=0.03,
                             map_width=mw, map_height=mh, trunc_depth_min=0.15, trunc_depth_max=5.05,
                             to_global=True, fill_value=0.0 if C else -np.inf)
    return proj, depth, pose, value

  jobs = [make(1, 64, 240, 320, 256, 256, 0), make(2, 9, 120, 160, 128, 128, 3),
          make(3, 1, 240, 320, 256, 256, 0), make(4, 33, 96, 128, 160, 160, 5)]

  def run(job, fused):
    proj, depth, pose, value = job
    if fused and value is None:
      return proj.orth_project_and_fuse(depth, cam_pose=pose)
    return proj.orth_project(depth, value_map=value, cam_pose=pose, get_height_map=value is not None)

  alone = [[t.clone() for t in run(j, f)] for j in jobs for f in (False, True)]
  torch.cuda.synchronize()
  errors = []

  def worker(tid):
    stream = torch.cuda.Stream()
    try:
      with torch.cuda.stream(stream):
        for it in range(40):
          k = (it * 3 + tid) % len(alone)
          out = run(jobs[k // 2], bool(k % 2))
          stream.synchronize()
          if not all(torch.equal(a, b) for a, b in zip(out, alone[k])):
            errors.append((tid, it, k))
            return
    except Exception as e:      # (a thread's exception would otherwise be lost)
      errors.append((tid, repr(e)))

  threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  assert not errors, errors


# --------------------------------------------------------------------------
# a slice of every mode of the builder-run parity campaign (tests/campaigns/parity_campaign.py)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("modes", [
    {},
    {"ONE_PITCH": "1", "CALLS": "1"},
    {"FILL_SPLIT": "1", "ONE_PITCH": "1", "CALLS": "1"},
    {"FLOW": "1"},
    {"FLOW": "1", "ONE_PITCH": "1", "SEMANTIC": "0"},
    {"FUSED": "1"},
    {"SUM": "1"},
    {"SUM": "mean"},
    {"ODD": "1"},
    {"OFFSETS": "1", "ONE_PITCH": "1", "CALLS": "1"},
    {"DC": "1", "CALLS": "1"},
    {"FINE": "1"},
], ids=lambda m: "+".join(f"{k}={v}" for k, v in m.items()) or "default")
def test_campaign_slice(dmap, oracle, modes):
  """Thirty seeded random configurations of each campaign mode against the oracle (cells and heights bit
  for bit; order-dependent sums 1e-5): the modes that found the defects of earlier rounds, plus this round's
  (where the fill duty runs, the projection + flow call with the one-kernel form switched at random)."""
  import importlib
  sys.path.insert(0, os.path.join(ROOT, "tests", "campaigns"))
  campaign = importlib.import_module("parity_campaign")
  campaign.configure({"DM_CAMPAIGN_" + k: v for k, v in modes.items()})
  try:
    bad = []
    for seed in range(4_000_000, 4_000_030):
      bm, bv, shape = campaign.one(seed)
      if bm or bv:
        bad.append((seed, shape, bm, bv))
    assert not bad, bad
  finally:
    campaign.reset_switches()
    campaign.configure({})
