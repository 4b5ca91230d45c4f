"""Multi-GPU: one process per GPU, frames sharded on the batch axis, partial
global maps fused with ONE element-wise max (or min) all-reduce.

The reference has no distributed code; this is the
north-star design: per-frame work needs no exchange, and because fusing maps
that share one frame is an element-wise max and the fill value is
the identity of that max, the only collective is
``all_reduce(op=MAX, float32, C*mh*mw)`` over RCCL (backend "nccl" on ROCm;
xGMI between the 8 GPUs of a node).  The mask is recomputed locally from the
reduced map (mask = f(map, fill)), so no mask exchange is needed.
The result is bit-identical for 1/2/4/8 ranks (max is exact and associative).
Partial maps of ``reduction='sum'`` are fused with ``ReduceOp.SUM`` (float32
tolerance, exact for point counts).
"""
from typing import Optional, Tuple

import torch

from .utils import Reduction


def shard_range(n_frames: int, rank: int, world_size: int) -> Tuple[int, int]:
  """Contiguous [start, stop) slice of the batch axis owned by ``rank``; the
  first ``n_frames % world_size`` ranks take one extra frame."""
  if not (0 <= rank < world_size):
    raise ValueError(f"rank {rank} outside world of {world_size}")
  base, extra = divmod(n_frames, world_size)
  start = rank * base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


def all_reduce_fused(fused: torch.Tensor, reduction=None, group=None,
                     fill_value: Optional[float] = None) -> torch.Tensor:
  """In-place all-reduce of this rank's partial global map (C, mh, mw).

  ``reduction`` max (default) / min: element-wise ``ReduceOp.MAX`` / ``MIN`` -- exact and
  associative, the result is bit-identical for any number of ranks.

  ``reduction`` sum ("for reduction=sum use ncclSum (tolerance)"): the reference's
  scatter-add reduces INTO a canvas pre-filled with ``fill_value`` (utils.py:470-477), so a rank's
  partial map is ``fill + sum of its points``; ``ReduceOp.SUM`` over R ranks counts the fill R
  times and ``(R - 1) * fill`` is taken off again.  Float32 sums depend on the order of the
  additions: equal to one process within 1e-5 relative, exactly equal for sums that are exact in
  float32 (point counts of one-hot value maps).  ``fill_value`` must be finite for a sum.

  mean / prod partial maps cannot be fused from the maps alone and are refused.  A single
  process (no initialised process group) is a no-op, so the same code runs on one GPU.
  """
  import math
  import torch.distributed as dist
  red = Reduction(reduction)
  if red not in (Reduction.max, Reduction.min, Reduction.sum):
    raise ValueError("partial maps can be fused across ranks for max / min (exactly) and sum "
                     f"(to float32 tolerance), not for {red.value}")
  fill = 0.0 if fill_value is None else float(fill_value)
  if red is Reduction.sum and not math.isfinite(fill):
    raise ValueError("a cross-rank sum needs a finite fill_value (the canvas value every rank's "
                     "partial sum starts from)")
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
    return fused
  if red is Reduction.sum:
    dist.all_reduce(fused, op=dist.ReduceOp.SUM, group=group)
    extra = (dist.get_world_size(group) - 1) * fill
    if extra != 0.0:
      fused.sub_(extra)
    return fused
  op = dist.ReduceOp.MAX if red is Reduction.max else dist.ReduceOp.MIN
  dist.all_reduce(fused, op=op, group=group)
  return fused


def partial_sum_map(maps: torch.Tensor, fill_value: Optional[float] = None) -> torch.Tensor:
  """A rank's partial global map for ``reduction='sum'`` from its per-frame maps (B, C, mh, mw):
  what scatter-adding all of the rank's points into ONE canvas pre-filled with ``fill_value``
  gives, ``fill + sum_b (maps[b] - fill)`` (float32, to tolerance)."""
  fill = 0.0 if fill_value is None else float(fill_value)
  total = maps.sum(dim=0)
  if fill != 0.0:
    total -= (maps.shape[0] - 1) * fill
  return total


def project_and_fuse_sharded(proj, depth_map, cam_pose, value_map=None, valid_map=None,
                             group=None, **kwargs):
  """Each rank projects ITS frames (already sharded by the caller) and the
  partial global maps are fused across ranks.

  Returns ``(topdown, mask, fused, fused_mask)``: the rank's per-frame maps and
  the job-wide fused map (identical on every rank for max / min; for sum equal
  to float32 tolerance).
  """
  import torch.distributed as dist
  from .functional import mask_from_map
  red = Reduction(kwargs.get("reduction", proj.reduction))
  fill = kwargs.get("fill_value", proj.fill_value)
  many = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
  if red is Reduction.sum:
    top, mask = proj.orth_project(depth_map, value_map=value_map, valid_map=valid_map,
                                  cam_pose=cam_pose, **kwargs)
    fused = partial_sum_map(top, fill)
    all_reduce_fused(fused, red, group, fill_value=fill)
    return top, mask, fused, mask_from_map(fused, fill)
  top, mask, fused, fmask = proj.orth_project_and_fuse(
      depth_map, value_map=value_map, valid_map=valid_map, cam_pose=cam_pose, **kwargs)
  before = fused.data_ptr()
  all_reduce_fused(fused, red, group)
  if many:
    fmask = mask_from_map(fused, fill)
  assert fused.data_ptr() == before
  return top, mask, fused, fmask
