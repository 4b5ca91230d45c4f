"""The N > 1 path on CPU: world_size-2 gloo processes shard a batch, build their
partial global maps, fuse them with parallel.all_reduce_fused and must end up
with exactly the single-process result (max is exact, so bit-identical).

No GPU here: each rank's partial map comes from the CPU oracle (the checker),
the collective and the sharding logic are the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(oracle):
  H, W, mh, mw = 48, 64, 96, 96
  cx, cy, fx, fy = oracle.camera_intrinsics(W, H, np.radians(70.))
  return dict(width_offset=mw / 2., height_offset=mh / 2., cam_pitch=np.radians(-20.),
              cam_height=0.88, map_res=0.1, map_width=mw, map_height=mh, focal_x=fx,
              focal_y=fy, center_x=cx, center_y=cy, trunc_depth_min=0.15,
              trunc_depth_max=5.05, to_global=True, fill_value=-np.inf)


def _inputs(B=7, H=48, W=64):
  g = torch.Generator().manual_seed(42)
  depth = torch.empty(B, 1, H, W).uniform_(0.1, 8.0, generator=g).numpy()
  pose = torch.empty(B, 3).uniform_(-1, 1, generator=g).numpy()
  return depth, pose


def _worker(rank, world, port, out_dir):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from oracle import oracle
  from dungeon_maps_amd import parallel
  depth, pose = _inputs()
  lo, hi = parallel.shard_range(len(depth), rank, world)
  part, _ = oracle.orth_project(depth[lo:hi], cam_pose=pose[lo:hi], fused=True, **_cfg(oracle))
  fused = torch.from_numpy(part.copy())
  parallel.all_reduce_fused(fused, "max")
  np.save(os.path.join(out_dir, f"fused_{rank}.npy"), fused.numpy())
  with pytest.raises(ValueError):
    parallel.all_reduce_fused(fused, "mean")
  with pytest.raises(ValueError):
    parallel.all_reduce_fused(fused, "sum", fill_value=-np.inf)     # a sum starts from a finite canvas
  # reduction='sum' (ncclSum, tolerance): point counts of a one-hot value map (exact in
  # float32) and sums of heights on top of a non-zero fill value
  rng = np.random.default_rng(3)
  labels = rng.integers(0, 3, size=depth.shape[:1] + depth.shape[2:])
  onehot = np.eye(3, dtype=np.float32)[labels].transpose(0, 3, 1, 2).copy()
  for tag, value, fill in (("counts", onehot, 0.0), ("heights", None, 1.5), ("heights01", None, 0.1)):
    kw = dict(_cfg(oracle), fill_value=fill, reduction="sum")
    maps, masks = oracle.orth_project(depth[lo:hi], value_map=None if value is None else value[lo:hi],
                                      cam_pose=pose[lo:hi], **kw)
    part, occ = parallel.partial_sum_map(torch.from_numpy(maps.copy()), torch.from_numpy(masks.copy()), fill)
    parallel.all_reduce_fused(part, "sum", fill_value=fill, occupied=occ)
    np.save(os.path.join(out_dir, f"sum_{tag}_{rank}.npy"), part.numpy())
    np.save(os.path.join(out_dir, f"occ_{tag}_{rank}.npy"), occ.numpy())
  dist.destroy_process_group()


def test_two_ranks_equal_one_rank(oracle, tmp_path):
  world = 2
  port = 29500 + os.getpid() % 2000
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  depth, pose = _inputs()
  want, wmask = oracle.orth_project(depth, cam_pose=pose, fused=True, **_cfg(oracle))
  for r in range(world):
    got = np.load(tmp_path / f"fused_{r}.npy")
    np.testing.assert_array_equal(got, want)
    # mask = f(map, fill): recomputed locally after the all-reduce
    np.testing.assert_array_equal(np.isfinite(got), wmask)
  # the cross-rank sum against ONE process scatter-adding every frame's points into one canvas
  # (the oracle's fused sum): counts exactly, heights to north_star's 1e-5
  rng = np.random.default_rng(3)
  labels = rng.integers(0, 3, size=depth.shape[:1] + depth.shape[2:])
  onehot = np.eye(3, dtype=np.float32)[labels].transpose(0, 3, 1, 2).copy()
  # (fill 0.1 is not exactly summable in float32: B * fill - (B - 1) * fill != fill -- emptiness must come
  # from the masks, never from a subtraction)
  for tag, value, fill in (("counts", onehot, 0.0), ("heights", None, 1.5), ("heights01", None, 0.1)):
    kw = dict(_cfg(oracle), fill_value=fill, reduction="sum")
    maps, masks = oracle.orth_project(depth, value_map=value, cam_pose=pose, **kw)
    hit = masks.any(0)
    one = np.where(hit, np.where(masks, maps.astype(np.float64) - np.float32(fill), 0.0).sum(0) + np.float32(fill),
                   np.float32(fill))
    for r in range(world):
      got = np.load(tmp_path / f"sum_{tag}_{r}.npy")
      occ = np.load(tmp_path / f"occ_{tag}_{r}.npy")
      np.testing.assert_array_equal(occ, hit)                       # the job-wide occupancy, on every rank
      np.testing.assert_array_equal(got[~hit], np.float32(fill))    # empty cells hold EXACTLY the fill value
      if tag == "counts":
        np.testing.assert_array_equal(got, one.astype(np.float32))
        assert got.max() >= 2                       # cells several frames of both ranks hit
      else:
        np.testing.assert_allclose(got, one, rtol=1e-5, atol=1e-5)
      # the mask of the fused map (mask_from_map: map - fill != 0) marks no empty cell
      assert not ((got - np.float32(fill)) != 0)[~hit].any()


def test_shard_range_partitions_the_batch():
  from dungeon_maps_amd import parallel
  for n in (0, 1, 7, 64, 513):
    for world in (1, 2, 3, 8):
      spans = [parallel.shard_range(n, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [b - a for a, b in spans]
      assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    parallel.shard_range(4, 2, 2)


def test_single_process_all_reduce_is_identity():
  from dungeon_maps_amd import parallel
  t = torch.randn(1, 4, 4)
  assert parallel.all_reduce_fused(t.clone(), None).equal(t)


def _gpu_worker(rank, world, port, out_dir):
  """One rank of the product's N > 1 path on a real GPU (both ranks share cuda:0; the collective
  runs over gloo because RCCL refuses two ranks on one device)."""
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  import dungeon_maps_amd as dmap
  from dungeon_maps_amd import parallel
  depth, pose = _inputs(B=9, H=96, W=128)
  lo, hi = parallel.shard_range(len(depth), rank, world)
  proj = dmap.MapProjector(width=128, height=96, hfov=np.radians(70.), cam_pitch=np.radians(-20.),
                           cam_height=0.88, width_offset=64., height_offset=64., map_res=0.05,
                           map_width=128, map_height=128, trunc_depth_min=0.15, trunc_depth_max=5.05,
                           to_global=True, fill_value=-np.inf)
  d = torch.from_numpy(depth[lo:hi]).cuda()
  top, mask, fused, fmask = parallel.project_and_fuse_sharded(proj, d, pose[lo:hi])
  torch.cuda.synchronize()
  np.save(os.path.join(out_dir, f"gpu_fused_{rank}.npy"), fused.cpu().numpy())
  np.save(os.path.join(out_dir, f"gpu_fmask_{rank}.npy"), fmask.cpu().numpy())
  np.save(os.path.join(out_dir, f"gpu_top_{rank}.npy"), top.cpu().numpy())
  dist.destroy_process_group()


@pytest.mark.gpu
def test_two_gpu_ranks_equal_one_rank(oracle, tmp_path):
  import torch
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU (run with -m gpu on an MI355X box)")
  world = 2
  port = 31500 + os.getpid() % 2000
  mp.spawn(_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  depth, pose = _inputs(B=9, H=96, W=128)
  cx, cy, fx, fy = oracle.camera_intrinsics(128, 96, np.radians(70.))
  kw = dict(width_offset=64., height_offset=64., cam_pitch=np.radians(-20.), cam_height=0.88,
            map_res=0.05, map_width=128, map_height=128, focal_x=fx, focal_y=fy, center_x=cx,
            center_y=cy, trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True,
            fill_value=-np.inf)
  want, wmask = oracle.orth_project(depth, cam_pose=pose, fused=True, **kw)
  per_frame, _ = oracle.orth_project(depth, cam_pose=pose, **kw)
  tops = []
  for r in range(world):
    np.testing.assert_array_equal(np.load(tmp_path / f"gpu_fused_{r}.npy"), want)
    np.testing.assert_array_equal(np.load(tmp_path / f"gpu_fmask_{r}.npy"), wmask)
    tops.append(np.load(tmp_path / f"gpu_top_{r}.npy"))
  np.testing.assert_array_equal(np.concatenate(tops), per_frame)     # the shards, in rank order
