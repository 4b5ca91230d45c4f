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


def all_reduce_fused(fused: torch.Tensor, reduction=None, group=None) -> torch.Tensor:
  """In-place all-reduce of this rank's partial global map (C, mh, mw).

  ``reduction`` is max (default) or min; other reductions are not exact under
  re-association and are refused.  A single process (no initialised process
  group) is a no-op, so the same code runs on one GPU.
  """
  import torch.distributed as dist
  red = Reduction(reduction)
  if red not in (Reduction.max, Reduction.min):
    raise ValueError("only max/min partial maps can be fused across ranks exactly")
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
    return fused
  op = dist.ReduceOp.MAX if red is Reduction.max else dist.ReduceOp.MIN
  dist.all_reduce(fused, op=op, group=group)
  return fused


def project_and_fuse_sharded(proj, depth_map, cam_pose, value_map=None, valid_map=None,
                             group=None, **kwargs):
  """Each rank projects ITS frames (already sharded by the caller) and the
  partial global maps are fused across ranks.

  Returns ``(topdown, mask, fused, fused_mask)``: the rank's per-frame maps and
  the job-wide fused map (identical on every rank).
  """
  from .functional import mask_from_map
  top, mask, fused, fmask = proj.orth_project_and_fuse(
      depth_map, value_map=value_map, valid_map=valid_map, cam_pose=cam_pose, **kwargs)
  red = kwargs.get("reduction", proj.reduction)
  before = fused.data_ptr()
  all_reduce_fused(fused, red, group)
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    fill = kwargs.get("fill_value", proj.fill_value)
    fmask = mask_from_map(fused, fill)
  assert fused.data_ptr() == before
  return top, mask, fused, fmask
