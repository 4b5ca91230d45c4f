"""Host-side camera state -> device ``dm_frame`` table.

The HIP kernels never evaluate sin/cos: the two Rodrigues matrices of every
frame are built here, on the CPU, in float32 with the reference's op order
(reference utils.py:303-327 via ``utils.rotation_matrix``) and shipped to the
library as one (B, 32) float32 host table (``dm_frame`` in
include/dungeon_maps_amd.h), which stages it to the GPU on the call's stream.
That is what makes the integer cell indices reproduce the reference's CPU path
bit for bit regardless of the device's libm.
"""
from typing import Optional

import torch

from . import utils
from ._native import FRAME_FLOATS

_AXIS_X = torch.tensor([[1., 0., 0.]])
_AXIS_Y = torch.tensor([[0., 1., 0.]])


def _column(value, batch: int, width: Optional[int] = None) -> torch.Tensor:
  """CPU float32 (batch,) or (batch, width) column from a scalar / list /
  tensor that has either 1 or ``batch`` rows."""
  t = utils.to_tensor(value).detach().to(device="cpu", dtype=torch.float32)
  t = t.reshape(-1) if width is None else t.reshape(-1, width)
  if t.shape[0] == 1 and batch != 1:
    t = t.expand(batch, *t.shape[1:])
  if t.shape[0] != batch:
    raise ValueError(f"per-frame argument has {t.shape[0]} rows, expected 1 or {batch}")
  return t


def build_frame_table(batch: int, cam_pose, cam_pitch, cam_height, width_offset,
                      height_offset) -> torch.Tensor:
  """(batch, 32) float32 CPU tensor laid out as ``dm_frame``:
  [0:9] Rp, [9] cam_height, [10:19] Ry, [19] tx, [20] tz, [21] woff, [22] hoff."""
  pose = _column([0., 0., 0.] if cam_pose is None else cam_pose, batch, 3)
  table = torch.zeros((batch, FRAME_FLOATS), dtype=torch.float32)
  table[:, 0:9] = utils.rotation_matrix(_AXIS_X, _column(cam_pitch, batch)).reshape(batch, 9)
  table[:, 9] = _column(cam_height, batch)
  table[:, 10:19] = utils.rotation_matrix(_AXIS_Y, pose[:, 2]).reshape(batch, 9)
  table[:, 19] = pose[:, 0]
  table[:, 20] = pose[:, 1]
  table[:, 21] = _column(width_offset, batch)
  table[:, 22] = _column(height_offset, batch)
  return table
