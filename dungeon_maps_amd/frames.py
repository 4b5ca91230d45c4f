"""Host-side camera state -> ``dm_frame`` table.

The HIP kernels never evaluate sin/cos: the two Rodrigues matrices of every
frame are built here, on the CPU, in float32, and handed to the library as one
(B, 32) float32 host table (``dm_frame`` in include/dungeon_maps_amd.h), which
stages it to the GPU on the call's stream.  That is what makes the integer cell
indices reproduce the reference's CPU path bit for bit regardless of the
device's libm.

The matrices are those of reference utils.py:303-327,
``R = (I + sin(a) S) + (1 - cos(a)) S^2`` evaluated element-wise in float32,
specialised to the two axes the projector uses (maps.py:790, 885):

  about X (pitch):  [[1, 0, 0], [0, d, -s], [0, s, d]]
  about Y (yaw):    [[d, 0, s], [0, 1, 0], [-s, 0, d]]      d = 1 - (1 - cos a)

with ``a`` clamped to 0 when |a| <= 1e-3 and sin/cos taken from torch's CPU
kernels (the libm the reference itself calls).  tests/test_host_api.py checks
this closed form against ``utils.rotation_matrix`` (the generic op-for-op
version) and against the reference-generated fixture.
"""
from typing import Optional

import numpy as np
import torch

from ._native import FRAME_FLOATS

_F32 = np.float32
_ONE = np.float32(1.0)
_ANGLE_EPS = np.float32(0.001)   # reference utils.py:47


def _is_scalar(v) -> bool:
  return isinstance(v, (int, float, np.floating, np.integer))


def _as_scalar(v):
  """A one-element CPU tensor or array as a python float; anything else unchanged."""
  if type(v) is torch.Tensor:
    if v.numel() == 1 and v.device.type == "cpu" and v.dtype is torch.float32 and not v.requires_grad:
      return v.item()
  elif type(v) is np.ndarray and v.size == 1 and v.dtype == _F32:
    return float(v.reshape(-1)[0])
  return v


def _column(value, batch: int, width: Optional[int] = None) -> np.ndarray:
  """float32 (batch,) or (batch, width) array from a scalar / list / array /
  tensor that has either 1 or ``batch`` rows."""
  if torch.is_tensor(value):
    if value.device.type != "cpu" or value.dtype != torch.float32 or value.requires_grad:
      value = value.detach().to(device="cpu", dtype=torch.float32)
    value = value.numpy()
  a = np.asarray(value, dtype=_F32)
  a = a.reshape(-1) if width is None else a.reshape(-1, width)
  if a.shape[0] == 1 and batch != 1:
    a = np.broadcast_to(a, (batch,) + a.shape[1:])
  if a.shape[0] != batch:
    raise ValueError(f"per-frame argument has {a.shape[0]} rows, expected 1 or {batch}")
  return a


def _sin_cos(angles: np.ndarray):
  """float32 sin/cos with torch's CPU kernels; |a| <= eps clamps to 0."""
  a = np.array(angles, dtype=_F32)            # private, contiguous copy
  small = np.abs(a) <= _ANGLE_EPS
  if small.any():
    a[small] = 0.0
  t = torch.from_numpy(a)
  return torch.sin(t).numpy(), torch.cos(t).numpy()


_scalar_terms = {}   # angle (python float) -> (sin, 1 - (1 - cos)) as float32


def _terms(angle, batch: int):
  """(s, d) of reference utils.py:326 for a scalar or per-frame angle."""
  if _is_scalar(angle):
    key = float(angle)
    hit = _scalar_terms.get(key)
    if hit is None:
      s, c = _sin_cos(np.array([key], dtype=_F32))
      hit = (s[0], _ONE - (_ONE - c[0]))
      if len(_scalar_terms) < 4096:
        _scalar_terms[key] = hit
    return hit
  s, c = _sin_cos(_column(angle, batch))
  return s, _ONE - (_ONE - c)            # (1 + sin*0) + (1 - cos) * (-1), float32


_static_tables = {}   # scalar camera state -> (batch, 32) table with everything but yaw / pose


def _static_columns(batch: int, cam_pitch, cam_height, width_offset, height_offset,
                    inverse_pitch: bool) -> np.ndarray:
  """The columns that do not depend on the pose; yaw = identity, no translation."""
  table = np.zeros((batch, FRAME_FLOATS), dtype=_F32)
  if inverse_pitch:
    neg = -cam_pitch if _is_scalar(cam_pitch) else -_column(cam_pitch, batch)
    si, di = _terms(neg, batch)
    table[:, 23] = _ONE
    table[:, 27] = di
    table[:, 28] = -si
    table[:, 30] = si
    table[:, 31] = di
  sp, dp = _terms(cam_pitch, batch)
  table[:, 0] = _ONE
  table[:, 4] = dp
  table[:, 5] = -sp
  table[:, 7] = sp
  table[:, 8] = dp
  table[:, 9] = cam_height if _is_scalar(cam_height) else _column(cam_height, batch)
  table[:, 10] = _ONE
  table[:, 14] = _ONE
  table[:, 18] = _ONE
  table[:, 21] = width_offset if _is_scalar(width_offset) else _column(width_offset, batch)
  table[:, 22] = height_offset if _is_scalar(height_offset) else _column(height_offset, batch)
  return table


_static_tensors = {}   # scalar camera state -> torch (batch, 32) table with everything but yaw / pose


def build_frame_table(batch: int, cam_pose, cam_pitch, cam_height, width_offset,
                      height_offset, inverse_pitch: bool = False) -> torch.Tensor:
  """(batch, 32) float32 CPU tensor laid out as ``dm_frame``:
  [0:9] Rp, [9] cam_height, [10:19] Ry, [19] tx, [20] tz, [21] woff, [22] hoff,
  [23:32] rotate(X, -pitch) when ``inverse_pitch`` (camera_affine_grid)."""
  # (one-element tensors / arrays count as scalars: MapBuilder hands the offsets over as 0-d tensors,
  # compute_center_offsets' return values -- float32 -> python float -> float32 is exact)
  cam_pitch, cam_height = _as_scalar(cam_pitch), _as_scalar(cam_height)
  width_offset, height_offset = _as_scalar(width_offset), _as_scalar(height_offset)
  scalar_rig = _is_scalar(cam_pitch) and _is_scalar(cam_height) and _is_scalar(width_offset) \
      and _is_scalar(height_offset)
  if scalar_rig and cam_pose is not None:
    # The usual call (one camera rig, shared offsets, a new set of poses): the columns that do not
    # change are cached per rig; torch's CPU sin / cos of the yaw column (the reference's own libm
    # calls) and ONE native call (dm_frames_fill_f32, host code) fill in the rest -- half the
    # host time of the numpy route below, same values.
    key = (batch, float(cam_pitch), float(cam_height), float(width_offset), float(height_offset),
           inverse_pitch)
    entry = _static_tensors.get(key)
    if entry is None:
      from . import _native
      static = torch.from_numpy(_static_columns(batch, cam_pitch, cam_height, width_offset,
                                                height_offset, inverse_pitch))
      entry = (static, static.data_ptr(), _native.lib().dm_frames_fill_f32)
      if len(_static_tensors) >= 256:      # (offsets that change from call to call: start over, stay bounded)
        _static_tensors.clear()
      _static_tensors[key] = entry
    pose = cam_pose
    if type(pose) is np.ndarray and pose.dtype == _F32 and pose.size == 3 * batch and pose.flags.c_contiguous:
      pose = torch.from_numpy(pose.reshape(batch, 3))        # (the demo's pose: a float32 (3,) array)
    elif not (type(pose) is torch.Tensor and pose.dtype is torch.float32 and pose.shape == (batch, 3)
              and pose.is_contiguous() and pose.device.type == "cpu" and not pose.requires_grad):
      pose = torch.from_numpy(np.ascontiguousarray(_column(cam_pose, batch, 3)))
    if batch == 1:
      # one frame: sin / cos of the whole (1, 3) record, the yaw's at float offset 2 -- no slicing call.  (Three
      # elements or one: torch's CPU kernel takes them through the same tail of its loop, so the yaw's sine is the
      # value sin(pose[:, 2]) gives; tests/test_host_api.py compares the two routes.)
      s, c = torch.sin(pose), torch.cos(pose)
      sp, cp = s.data_ptr() + 8, c.data_ptr() + 8
    else:
      yaw = pose[:, 2]
      s, c = torch.sin(yaw), torch.cos(yaw)
      sp, cp = s.data_ptr(), c.data_ptr()
    # (a fresh table per call: the library reads it inside the call, but a prepared batch keeps it)
    table = torch.empty((batch, FRAME_FLOATS), dtype=torch.float32)
    if entry[2](entry[1], batch, pose.data_ptr(), sp, cp, table.data_ptr()) != 0:
      raise ValueError("dm_frames_fill_f32 refused its arguments")
    return table
  if scalar_rig:
    key = (batch, float(cam_pitch), float(cam_height), float(width_offset),
           float(height_offset), inverse_pitch)
    static = _static_tables.get(key)
    if static is None:
      static = _static_columns(batch, cam_pitch, cam_height, width_offset, height_offset,
                               inverse_pitch)
      if len(_static_tables) < 256:
        _static_tables[key] = static
    table = static.copy()
  else:
    table = _static_columns(batch, cam_pitch, cam_height, width_offset, height_offset,
                            inverse_pitch)
  if cam_pose is not None:
    pose = _column(cam_pose, batch, 3)
    sy, dy = _terms(pose[:, 2], batch)
    table[:, 10] = dy
    table[:, 12] = sy
    table[:, 16] = -sy
    table[:, 18] = dy
    table[:, 19:21] = pose[:, 0:2]
  return torch.from_numpy(table)
