"""Tensor utilities with the reference's names and semantics.

API mirror of reference ``dungeon_maps/utils.py`` (``__all__`` at utils.py:21-42)
so user code that does ``dmap.utils.<name>`` keeps working.  Everything here is
host-side glue or small-tensor math; the heavy lifting of the projector is in
the HIP library (``_native.py``).  ``scatter_tensor`` -- the reference's only
third-party native call site (utils.py:475-477) -- is served by the HIP
library as well and has no CPU implementation.
"""
import enum
from dataclasses import dataclass
from typing import Any, Optional, Sequence, Tuple, Union

import numpy as np
import torch

__all__ = [
    "NINF", "Reduction", "CameraIntrinsics", "get_camera_intrinsics",
    "to_numpy", "to_tensor", "to_tensor_like", "validate_tensors",
    "translate", "rotate", "rotation_matrix", "ravel_index", "scatter_tensor",
    "to_4D_image", "from_4D_image", "generate_image_coords",
    "generate_crop_grid", "image_sample",
]

NINF = -np.inf
ANGLE_EPS = 0.001  # reference utils.py:47


class Reduction(str, enum.Enum):
  """Reduction of values that fall into one cell (reference utils.py:52-67)."""
  max = "max"
  min = "min"
  sum = "sum"
  mean = "mean"
  prod = "prod"

  @classmethod
  def _missing_(cls, value):
    return cls.max if value is None else None


@dataclass
class CameraIntrinsics:
  cx: float
  cy: float
  fx: float
  fy: float


def get_camera_intrinsics(width, height, hfov, vfov=None) -> CameraIntrinsics:
  """Pinhole intrinsics in float64, as reference utils.py:94-116: principal
  point at the image centre, fy = fx unless a vertical fov is given."""
  half_w, half_h = width / 2., height / 2.
  fx = half_w / np.tan(hfov / 2.)
  fy = fx if vfov is None else half_h / np.tan(vfov / 2.)
  return CameraIntrinsics(cx=half_w, cy=half_h, fx=fx, fy=fy)


# ----------------------------------------------------------------- conversions
def to_numpy(inputs: Any, dtype=None) -> np.ndarray:
  arr = inputs.detach().cpu().numpy() if torch.is_tensor(inputs) else np.asarray(inputs)
  return arr.astype(dtype or arr.dtype)


def to_tensor(inputs: Any, dtype=None, device=None, **kwargs) -> torch.Tensor:
  if not torch.is_tensor(inputs):
    inputs = (torch.from_numpy(inputs) if isinstance(inputs, np.ndarray)
              else torch.tensor(inputs, dtype=dtype))
  return inputs.to(device=device, dtype=dtype, **kwargs)


def to_tensor_like(inputs: Any, tensor: torch.Tensor) -> torch.Tensor:
  assert torch.is_tensor(tensor), f"`tensor` must be a torch.Tensor, got {type(tensor)}"
  return to_tensor(inputs, dtype=tensor.dtype, device=tensor.device)


def validate_tensors(*args, same_device=None, same_dtype=None, keep_tuple=False):
  """Convert every argument to a tensor; ``same_device`` / ``same_dtype`` may be
  True (= take it from the first argument) or an explicit device / dtype.
  (The reference's ``same_dtype=True`` branch is broken, utils.py:215-217; here
  it does what its docstring says.)"""
  if not args:
    return None
  first = to_tensor(args[0])
  device = first.device if same_device is True else (same_device or None)
  dtype = first.dtype if same_dtype is True else (same_dtype or None)
  out = tuple(to_tensor(a, device=device, dtype=dtype) for a in args)
  return out[0] if len(out) == 1 and not keep_tuple else out


# ------------------------------------------------------------------ 3-D affine
def translate(points: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
  """points (b, ..., 3) + offsets (b, 3)."""
  points, offsets = validate_tensors(points, offsets, same_device=True,
                                     same_dtype=torch.float32)
  b = points.shape[0]
  return (points.reshape(b, -1, 3) + offsets.reshape(-1, 1, 3)).reshape(points.shape)


def rotation_matrix(axis, angle, angle_eps: float = ANGLE_EPS) -> torch.Tensor:
  """Rodrigues matrices R = I + sin(a) S + (1 - cos(a)) S^2, shape (b, 3, 3).

  Follows the float32 op order of reference utils.py:303-327 exactly (this is
  what pins the integer cell indices), but broadcasts a single ``axis`` over
  the batch -- the reference crashes there for b > 1 (utils.py:311-316).
  |angle| <= angle_eps is clamped to 0.
  """
  axis, angle = validate_tensors(axis, angle, same_device=True, same_dtype=torch.float32)
  angle = angle.reshape(-1, 1)
  b = angle.shape[0]
  axis = axis.reshape(-1, 3).expand(b, 3)
  unit = axis / torch.linalg.norm(axis, dim=-1, keepdim=True)
  ux, uy, uz = unit.unbind(-1)
  o = torch.zeros_like(ux)
  skew = torch.stack((o, -uz, uy, uz, o, -ux, -uy, ux, o), dim=-1)   # (b, 9)
  skew2 = torch.bmm(skew.view(b, 3, 3), skew.view(b, 3, 3)).reshape(b, 9)
  angle = torch.where(angle.abs() > angle_eps, angle, torch.zeros_like(angle))
  eye = torch.eye(3, device=angle.device).reshape(1, 9)
  rot = eye + torch.sin(angle) * skew + (1 - torch.cos(angle)) * skew2
  return rot.view(b, 3, 3)


def rotate(points, axis, angle, angle_eps: float = ANGLE_EPS) -> torch.Tensor:
  """Rotate points (b, ..., 3) about ``axis`` by ``angle`` (b,):
  out_i = sum_j R[j, i] p_j  (reference utils.py:329)."""
  points = validate_tensors(points, same_dtype=torch.float32)
  rot = rotation_matrix(to_tensor(axis, device=points.device),
                        to_tensor(angle, device=points.device), angle_eps)
  b = points.shape[0]
  if rot.shape[0] != b:
    rot = rot.expand(b, 3, 3)
  # (the contraction over the FIRST index of R as ONE einsum, the call the reference makes,
  # utils.py:329: which BLAS kernel torch's CPU matmul picks -- and with it whether the three
  # products are summed as an FMA chain -- depends on the shape, the call and the machine, so CPU
  # tensors get the reference's own bits only through the reference's own call.  GPU tensors go
  # through dm_affine_points_f32, always the FMA chain.)
  return torch.einsum("bji,b...j->b...i", rot, points)


# ------------------------------------------------------------------- indexing
def ravel_index(index, shape: Sequence[int], keepdim: bool = False) -> torch.Tensor:
  """np.ravel_multi_index for (..., n) index tensors, e.g.
  ravel_index([[3, 2, 3], [0, 2, 1]], (6, 5, 4)) -> [71, 9]."""
  index = to_tensor(index).to(torch.int64)
  strides = np.ones(len(shape), dtype=np.int64)
  strides[:-1] = np.cumprod(np.asarray(shape[:0:-1], dtype=np.int64))[::-1]
  weights = torch.from_numpy(strides).to(index.device)
  return (index * weights).sum(dim=-1, keepdim=keepdim)


def scatter_tensor(canvas, indices, values, masks=None, fill_value=None,
                   reduction=None, _validate_args: bool = True
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
  """Scatter ``values`` (b..., N) into ``canvas`` (b..., d1..dn) at ``indices``
  (b..., N, n) with a reduction; returns (canvas, changed-mask).

  Semantics of reference utils.py:389-492: out-of-range / masked points are
  dropped, the canvas is first filled with ``fill_value`` if given and its
  content takes part in the reduction, and the mask is True where the canvas
  value changed.  Runs on the HIP library (dm_scatter_f32).
  """
  from . import functional
  return functional.scatter_nd(canvas, indices, values, masks, fill_value, reduction)


# --------------------------------------------------------------------- images
def to_4D_image(image: torch.Tensor) -> torch.Tensor:
  """(h,w) / (c,h,w) / (b,c,h,w) -> (b,c,h,w)."""
  nd = image.dim()
  assert nd in (2, 3, 4), f"only supports 2/3/4D images while {nd}-D are given."
  return image[(None,) * (4 - nd)]


def from_4D_image(image: torch.Tensor, ndims: int) -> torch.Tensor:
  assert image.dim() == 4, f"`image` must be a 4D tensor, while {image.dim()}-D are given."
  return image[(0,) * max(0, 4 - ndims)] if ndims < 4 else image


def generate_image_coords(image_shape, dtype=None, device=None):
  """(x, y) pixel coordinate grids broadcast to ``image_shape`` (..., h, w)."""
  if len(image_shape) < 2:
    raise ValueError(f"rank of `image_shape` must be at east 2D, got {len(image_shape)}")
  dtype = dtype or torch.float32
  h, w = image_shape[-2], image_shape[-1]
  xs = torch.arange(w, dtype=dtype, device=device).expand(*image_shape)
  ys = torch.arange(h, dtype=dtype, device=device).unsqueeze(-1).expand(*image_shape)
  return xs, ys


def generate_crop_grid(center, image_width, image_height, crop_width, crop_height,
                       device=None) -> torch.Tensor:
  """Normalised sampling grid (b, crop_h, crop_w, 2) of a crop window around
  ``center`` (b, 2) over an image that will be padded by one pixel per side
  (reference utils.py:571-611)."""
  center = to_tensor(center, device=device).reshape(-1, 2).to(torch.float32) + 1
  b = center.shape[0]
  pw, ph = image_width + 2, image_height + 2
  xs, ys = generate_image_coords((b, crop_height, crop_width), torch.float32, center.device)
  ox = (center[:, 0] - pw / 2.).view(-1, 1, 1)
  oy = (center[:, 1] - ph / 2.).view(-1, 1, 1)
  gx = (xs - crop_width / 2. + ox) / (pw / 2.)
  gy = (ys - crop_height / 2. + oy) / (ph / 2.)
  return torch.stack((gx, gy), dim=-1)


def image_sample(image, grid, fill_value=None, mode: str = "nearest",
                 _validate_args: bool = True) -> torch.Tensor:
  """grid_sample of the 1-px constant-padded image (reference utils.py:613-652)."""
  if _validate_args:
    image, grid = validate_tensors(image, grid, same_device=True)
    image = to_4D_image(image)
  border = "zeros" if fill_value is None else "border"
  padded = torch.nn.functional.pad(image, [1, 1, 1, 1], mode="constant",
                                   value=0.0 if fill_value is None else fill_value)
  out = torch.nn.functional.grid_sample(padded.to(grid.dtype), grid, mode=mode,
                                        padding_mode=border, align_corners=True)
  return out.to(image.dtype)
