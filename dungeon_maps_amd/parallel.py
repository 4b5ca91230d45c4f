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
                     fill_value: Optional[float] = None,
                     occupied: Optional[torch.Tensor] = None) -> torch.Tensor:
  """In-place all-reduce of this rank's partial global map (C, mh, mw).

  ``reduction`` max (default) / min: element-wise ``ReduceOp.MAX`` / ``MIN`` -- exact and
  associative, the result is bit-identical for any number of ranks.

  ``reduction`` sum ("for reduction=sum use ncclSum (tolerance)"): ``fused`` holds what
  this rank's points ADD to the canvas -- ``partial_sum_map``: the sum over its frames of
  ``map - fill`` where a frame's mask is set, zero elsewhere -- and ``occupied`` (bool, same shape)
  the cells any of its frames hit.  The sums are all-reduced with ``ReduceOp.SUM``, ``occupied`` with
  ``ReduceOp.MAX``; afterwards ``fused = fill + sum`` where a cell is occupied on any rank and
  EXACTLY ``fill`` elsewhere -- emptiness is never derived from a float32 subtraction (a fill such
  as 0.1 does not survive ``B * fill - (B - 1) * fill``).  Float32 sums depend on the order of the
  additions: equal to one process within 1e-5 relative, exactly equal for sums that are exact in
  float32 (point counts of one-hot value maps).  ``fill_value`` must be finite for a sum.

  mean / prod partial maps cannot be fused from the maps alone and are refused.  A single
  process (no initialised process group) skips the collectives, so the same code runs on one GPU.
  """
  import math
  import torch.distributed as dist
  red = Reduction(reduction)
  if red not in (Reduction.max, Reduction.min, Reduction.sum):
    raise ValueError("partial maps can be fused across ranks for max / min (exactly) and sum "
                     f"(to float32 tolerance), not for {red.value}")
  fill = 0.0 if fill_value is None else float(fill_value)
  if red is Reduction.sum:
    if not math.isfinite(fill):
      raise ValueError("a cross-rank sum needs a finite fill_value (the canvas value every rank's "
                       "partial sum starts from)")
    if occupied is None or occupied.shape != fused.shape:
      raise ValueError("a cross-rank sum needs `occupied`, the cells this rank's frames hit "
                       "(partial_sum_map returns it)")
  many = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
  if red is Reduction.sum:
    occ = occupied.to(torch.uint8)
    if many:
      dist.all_reduce(fused, op=dist.ReduceOp.SUM, group=group)
      dist.all_reduce(occ, op=dist.ReduceOp.MAX, group=group)
    fused.add_(fill)
    fused.masked_fill_(occ == 0, fill)
    occupied.copy_(occ.to(torch.bool))
    return fused
  if not many:
    return fused
  op = dist.ReduceOp.MAX if red is Reduction.max else dist.ReduceOp.MIN
  dist.all_reduce(fused, op=op, group=group)
  return fused


def partial_sum_map(maps: torch.Tensor, masks: torch.Tensor,
                    fill_value: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
  """A rank's contribution to the job-wide ``reduction='sum'`` map from its per-frame maps and masks
  (B, C, mh, mw): ``(added, occupied)`` with ``added = sum_b where(masks[b], maps[b] - fill, 0)`` --
  what the rank's points add on top of the canvas's fill value (utils.py:470-477 scatter-adds INTO
  the filled canvas), float32, to tolerance -- and ``occupied = any_b masks[b]``."""
  fill = 0.0 if fill_value is None else float(fill_value)
  added = torch.where(masks, maps - fill, torch.zeros((), dtype=maps.dtype, device=maps.device)).sum(dim=0)
  return added, masks.any(dim=0)


def project_and_fuse_sharded(proj, depth_map, cam_pose, value_map=None, valid_map=None,
                             group=None, **kwargs):
  """Each rank projects ITS frames (already sharded by the caller) and the
  partial global maps are fused across ranks.

  Returns ``(topdown, mask, fused, fused_mask)``: the rank's per-frame maps and
  the job-wide fused map (identical on every rank for max / min; for sum equal
  to float32 tolerance).
  """
  import math
  import torch.distributed as dist
  from .functional import mask_from_map
  red = Reduction(kwargs.get("reduction", proj.reduction))
  fill = kwargs.get("fill_value", proj.fill_value)
  # (refused up front, before anything is projected)
  if red not in (Reduction.max, Reduction.min, Reduction.sum):
    raise ValueError("partial maps can be fused across ranks for max / min (exactly) and sum "
                     f"(to float32 tolerance), not for {red.value}")
  if red is Reduction.sum:
    if not math.isfinite(0.0 if fill is None else float(fill)):
      raise ValueError("a cross-rank sum needs a finite fill_value (the canvas value every rank's "
                       "partial sum starts from)")
    if kwargs.get("get_height_map"):
      raise ValueError("project_and_fuse_sharded(reduction='sum') returns no height map: project the "
                       "heights with a second call")
  many = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
  if red is Reduction.sum:
    top, mask = proj.orth_project(depth_map, value_map=value_map, valid_map=valid_map,
                                  cam_pose=cam_pose, **kwargs)
    fused, occupied = partial_sum_map(top, mask, fill)
    all_reduce_fused(fused, red, group, fill_value=fill, occupied=occupied)
    return top, mask, fused, mask_from_map(fused, fill)
  top, mask, fused, fmask = proj.orth_project_and_fuse(
      depth_map, value_map=value_map, valid_map=valid_map, cam_pose=cam_pose, **kwargs)
  before = fused.data_ptr()
  all_reduce_fused(fused, red, group)
  if many:
    fmask = mask_from_map(fused, fill)
  assert fused.data_ptr() == before
  return top, mask, fused, fmask
