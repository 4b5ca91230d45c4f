"""Raw functional API (reference dungeon_maps/maps.py:121-1248), MI355X-native.

``orth_project`` -- the hot path -- and ``project`` / ``scatter_nd`` run on the
hand-written HIP library through the C ABI (include/dungeon_maps_amd.h).  The
remaining functions are the small point-set helpers callers use around it
(coordinate queries, offsets); they are thin float32 tensor formulas.

Differences from the reference, on purpose:
* batches work (the reference crashes for B > 1 in utils.rotate, utils.py:303-316);
  per-frame arguments may have 1 or B rows.
* inputs on the CPU are accepted, computed on the GPU and returned on the CPU;
  without a GPU the call raises -- there is no CPU fallback.
"""
import ctypes
from typing import Any, Optional, Tuple, Union

import numpy as np
import torch

from . import _native, frames, utils
from .utils import NINF, Reduction

__all__ = [
    "CenterMode", "get", "orth_project", "orth_project_and_fuse", "orth_project_and_flow", "orth_project_fused", "fuse_batch", "mask_from_map", "camera_affine_grid",
    "PreparedProjection", "prepare_orth_project",
    "depth_map_to_point_cloud", "height_map_to_point_cloud", "image_to_camera_space",
    "camera_to_image_space", "camera_to_local_space", "local_to_camera_space",
    "local_to_global_space", "global_to_local_space", "map_quantize",
    "map_dequantize", "project", "compute_center_offsets",
]

import enum
import threading


class CenterMode(str, enum.Enum):
  """Where the map is centred (reference maps.py:26-39)."""
  none = "none"
  origin = "origin"
  camera = "camera"

  @classmethod
  def _missing_(cls, value):
    return cls.none if value is None else None


def get(*args: Any) -> Any:
  """First argument that is not None (the last one if all are)."""
  for arg in args:
    if arg is not None:
      return arg
  return args[-1] if args else None


_REDUCTION_CODE = {
    Reduction.max: _native.REDUCE_MAX, Reduction.min: _native.REDUCE_MIN,
    Reduction.sum: _native.REDUCE_SUM, Reduction.mean: _native.REDUCE_MEAN,
    Reduction.prod: _native.REDUCE_PROD,
}


def _reduction_code(reduction) -> int:
  red = Reduction(reduction)
  if red not in _REDUCTION_CODE:
    raise ValueError(f"Invalid reduction method: {reduction}")
  return _REDUCTION_CODE[red]


def _compute_device(target: torch.device) -> torch.device:
  """The GPU the kernels run on: the tensors' own device when that is a GPU,
  else the current GPU.  No GPU -> error (no CPU path by design)."""
  if target.type == "cuda":
    # (with an index: device='cuda' names the current GPU, and a caller's tensors on cuda:0 compare unequal
    # to torch.device('cuda'))
    return target if target.index is not None else torch.device("cuda", torch.cuda.current_device())
  if not torch.cuda.is_available():
    raise RuntimeError(
        "dungeon_maps_amd computes on an AMD GPU through its HIP library and has no CPU "
        "fallback; no GPU is visible to PyTorch in this process.")
  return torch.device("cuda", torch.cuda.current_device())


class _NoSwitch:
  def __enter__(self):
    return None

  def __exit__(self, *exc):
    return False


_NO_SWITCH = _NoSwitch()


def _on_device(device: torch.device):
  """``torch.cuda.device(device)``, skipped when that GPU is current already (the switch
  costs ~2 us of the ~40 a projection call takes on the host)."""
  index = device.index
  if index is None or index == torch.cuda.current_device():
    return _NO_SWITCH
  return torch.cuda.device(device)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device: torch.device) -> int:
  """hipStream_t of PyTorch's current stream on ``device``, as an integer."""
  if _raw_stream is not None:       # the pointer itself; a Stream object costs ~4 us to build
    index = device.index
    return _raw_stream(torch.cuda.current_device() if index is None else index)
  return torch.cuda.current_stream(device).cuda_stream


def _image(x, device, dtype) -> torch.Tensor:
  t = x if torch.is_tensor(x) else utils.to_tensor(x)
  if t.dim() != 4:
    t = utils.to_4D_image(t)
  if t.device != device or t.dtype != dtype:
    t = t.to(device=device, dtype=dtype)
  return t if t.is_contiguous() else t.contiguous()


class _Call:
  """Canonicalised arguments of one orth_project call (maps.py:225-257)."""

  def __init__(self, depth_map, value_map, valid_map, cam_pose, width_offset,
               height_offset, cam_pitch, cam_height, map_res, map_width, map_height,
               focal_x, focal_y, center_x, center_y, trunc_depth_min, trunc_depth_max,
               trunc_height_max, clip_border, to_global, flip_h, fill_value, reduction,
               device):
    first = depth_map if torch.is_tensor(depth_map) else utils.to_tensor(depth_map)
    self.target = torch.device(device) if device is not None else first.device
    self.dev = _compute_device(self.target)
    self.depth = _image(first, self.dev, torch.float32)
    B, dc, H, W = self.depth.shape
    self.value = None if value_map is None else _image(value_map, self.dev, torch.float32)
    self.valid = None if valid_map is None else _image(valid_map, self.dev, torch.bool)
    vc = 0
    if self.value is not None:
      if self.value.shape[0] != B or self.value.shape[2:] != (H, W):
        raise ValueError(f"value_map {tuple(self.value.shape)} does not match depth_map "
                         f"{tuple(self.depth.shape)}")
      vc = self.value.shape[1]
      if dc not in (1, vc):
        raise ValueError(f"depth_map has {dc} channels; expected 1 or {vc} (value_map's)")
    valid_c = 0
    if self.valid is not None:
      # (a (b, C, h, w) valid map over a 1-channel depth map is refused like the reference
      # refuses it: maps.py:1155-1158 raises on the channel mismatch)
      if self.valid.shape[2:] != (H, W) or self.valid.shape[0] not in (1, B) \
          or self.valid.shape[1] not in (1, dc):
        raise ValueError(f"valid_map {tuple(self.valid.shape)} does not broadcast to "
                         f"depth_map {tuple(self.depth.shape)}")
      if self.valid.shape[0] != B:
        self.valid = self.valid.expand(B, -1, -1, -1).contiguous()
      valid_c = self.valid.shape[1]
    key = (B, dc, vc, H, W, int(map_height), int(map_width),
           int(clip_border) if clip_border is not None else 0, bool(flip_h), bool(to_global),
           _reduction_code(reduction), trunc_depth_min, trunc_depth_max, trunc_height_max,
           valid_c, float(center_x), float(center_y), float(focal_x), float(focal_y),
           float(map_res), fill_value)
    cached = _PARAMS_CACHE.get(key)
    if cached is None:
      cached = _make_params(key)
      if len(_PARAMS_CACHE) < 256:
        _PARAMS_CACHE[key] = cached
    self.params, self.ws_bytes = cached
    self.oc = vc if vc else dc
    # host-side dm_frame table; the library stages it to the GPU inside the call
    self.frames = frames.build_frame_table(B, cam_pose, cam_pitch, cam_height, width_offset,
                                           height_offset)

  def workspace(self) -> Tuple[Optional[torch.Tensor], int]:
    if self.ws_bytes == 0:
      return None, 0
    if self.ws_bytes > _WS_KEEP_BYTES or torch.cuda.is_current_stream_capturing():    # (a graph keeps its own)
      return torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev), self.ws_bytes
    # Small workspaces (few frames: the host-bound calls) are kept per (host thread, GPU, stream) and handed to
    # that thread's next call on that stream: the kernels of two calls one thread makes on one stream run in
    # order, and the workspace holds nothing between calls.  (Per thread: two threads on ONE stream enqueue their
    # launch sequences interleaved.  A stream's address can come back for a new stream after the old one is
    # destroyed: also then all earlier work on it has finished.)
    kept = getattr(_WS_LOCAL, "kept", None)
    if kept is None:
      kept = _WS_LOCAL.kept = {}
    key = (self.dev.index, _stream_ptr(self.dev))
    ws = kept.get(key)
    if ws is None or ws.numel() < self.ws_bytes:
      if len(kept) >= 16:
        kept.clear()
      ws = kept[key] = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev)
    return ws, self.ws_bytes


_WS_KEEP_BYTES = 16 << 20
_WS_LOCAL = threading.local()     # .kept: (GPU index, stream) -> uint8 workspace of this thread's small calls
_PARAMS_CACHE = {}   # call signature -> (dm_params, workspace bytes); both immutable afterwards


def _make_params(key):
  import ctypes
  (B, dc, vc, H, W, mh, mw, clip, flip_h, to_global, red, dmin, dmax, hmax, valid_c, cx, cy,
   fx, fy, res, fill) = key
  p = _native.Params()
  p.B, p.dc, p.vc, p.H, p.W = B, dc, vc, H, W
  p.mh, p.mw = mh, mw
  p.clip_border = clip
  p.flip_h = int(flip_h)
  p.to_global = int(to_global)
  p.reduction = red
  p.has_dmin = int(dmin is not None)
  p.has_dmax = int(dmax is not None)
  p.has_hmax = int(hmax is not None)
  p.valid_c = valid_c
  p.cx, p.cy, p.fx, p.fy = cx, cy, fx, fy
  p.res = res
  # fill_value None: the reference scatters into an all-zero canvas (maps.py:320)
  p.fill = 0.0 if fill is None else float(fill)
  p.dmin = float(dmin) if dmin is not None else 0.0
  p.dmax = float(dmax) if dmax is not None else 0.0
  p.hmax = float(hmax) if hmax is not None else 0.0
  need = _native.lib().dm_orth_project_workspace_bytes(ctypes.byref(p))
  return p, int(need)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
  return None if t is None else t.data_ptr()


def orth_project(
    depth_map, value_map, valid_map, cam_pose, width_offset, height_offset, cam_pitch,
    cam_height, map_res: float, map_width: int, map_height: int, focal_x: float,
    focal_y: float, center_x: float, center_y: float, trunc_depth_min: Optional[float],
    trunc_depth_max: Optional[float], trunc_height_max: Optional[float],
    clip_border: Optional[int], to_global: bool, flip_h: bool = True,
    fill_value: Optional[float] = None, reduction: Optional[Reduction] = None,
    get_height_map: bool = False, device: Optional[torch.device] = None,
    _validate_args: bool = True
) -> Union[Tuple[torch.Tensor, torch.Tensor],
           Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
  """Orthographic projection of depth maps (b, c, h, w) to top-down maps.

  Same signature, argument meaning and returns as reference maps.py:127-351:
  ``(topdown (b,C,mh,mw) f32, mask (b,C,mh,mw) bool[, height_map])``.  Without
  ``value_map`` the heights are projected and ``height_map`` is the very same
  tensor as ``topdown``; with a ``value_map`` the height map is a second
  NINF/max projection broadcast over the value channels (stride 0).

  One fused HIP launch sequence replaces the reference's ~230 ATen calls:
  unproject, pitch/yaw/translate, truncations, quantisation, scatter-reduce and
  the changed-mask (dm_orth_project_f32).
  """
  return _orth_project(depth_map, value_map, valid_map, cam_pose, width_offset, height_offset,
                       cam_pitch, cam_height, map_res, map_width, map_height, focal_x, focal_y,
                       center_x, center_y, trunc_depth_min, trunc_depth_max, trunc_height_max,
                       clip_border, to_global, flip_h, fill_value, reduction, get_height_map,
                       device, False, None, None)


def _orth_project(depth_map, value_map, valid_map, cam_pose, width_offset, height_offset,
                  cam_pitch, cam_height, map_res, map_width, map_height, focal_x, focal_y,
                  center_x, center_y, trunc_depth_min, trunc_depth_max, trunc_height_max,
                  clip_border, to_global, flip_h, fill_value, reduction, get_height_map, device,
                  _fuse, _fused_out, _out):
  """orth_project / orth_project_and_fuse behind one native call."""
  import ctypes
  call = _Call(depth_map, value_map, valid_map, cam_pose, width_offset, height_offset,
               cam_pitch, cam_height, map_res, map_width, map_height, focal_x, focal_y,
               center_x, center_y, trunc_depth_min, trunc_depth_max, trunc_height_max,
               clip_border, to_global, flip_h, fill_value, reduction, device)
  p = call.params
  shape = (p.B, call.oc, p.mh, p.mw)
  if _out is not None:
    topdown, mask = _out
    for t, dt in ((topdown, torch.float32), (mask, torch.bool)):
      if tuple(t.shape) != shape or t.dtype != dt or t.device != call.dev or not t.is_contiguous():
        raise ValueError(f"`out` must be contiguous {dt} {shape} tensors on {call.dev}")
  else:
    topdown = torch.empty(shape, dtype=torch.float32, device=call.dev)
    mask = torch.empty(shape, dtype=torch.bool, device=call.dev)
  height = None
  if get_height_map and p.vc:
    height = torch.empty((p.B, p.dc, p.mh, p.mw), dtype=torch.float32, device=call.dev)
  fused = fmask = None
  if _fuse and _fused_out is not None:
    fused, fmask = _fused_out
    for t, dt in ((fused, torch.float32), (fmask, torch.bool)):
      if tuple(t.shape) != shape[1:] or t.dtype != dt or t.device != call.dev \
          or not t.is_contiguous():
        raise ValueError(f"`fused_out` must be contiguous {dt} {shape[1:]} tensors on {call.dev}")
  elif _fuse:
    fused = torch.empty(shape[1:], dtype=torch.float32, device=call.dev)
    fmask = torch.empty(shape[1:], dtype=torch.bool, device=call.dev)
  ws, ws_bytes = call.workspace()
  _native.check_status()
  with _on_device(call.dev):
    _native.check(_native.lib().dm_orth_project_f32(
        ctypes.byref(p), _ptr(call.frames), _ptr(call.depth), _ptr(call.value),
        _ptr(call.valid), _ptr(topdown), _ptr(mask), _ptr(height), _ptr(fused), _ptr(fmask),
        _ptr(ws), ws_bytes, _native.status_ptr(), _stream_ptr(call.dev)))
  if call.target != call.dev:
    topdown, mask = topdown.to(call.target), mask.to(call.target)
    height = None if height is None else height.to(call.target)
    if _fuse:
      fused, fmask = fused.to(call.target), fmask.to(call.target)
  if _fuse:
    return topdown, mask, fused, fmask
  if not get_height_map:
    return topdown, mask
  if height is None:
    return topdown, mask, topdown
  return topdown, mask, torch.broadcast_to(height, topdown.shape)


def orth_project_and_fuse(depth_map, value_map, valid_map, cam_pose, width_offset,
                          height_offset, cam_pitch, cam_height, map_res, map_width, map_height,
                          focal_x, focal_y, center_x, center_y, trunc_depth_min,
                          trunc_depth_max, trunc_height_max, clip_border, to_global=True,
                          flip_h=True, fill_value=NINF, reduction=None, device=None,
                          fused_out=None, out=None
                          ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
  """``orth_project`` plus the batch-fused map in the same launch sequence:
  returns ``(topdown (b,C,mh,mw), mask, fused (C,mh,mw), fused_mask)`` with
  ``fused = max (or min) over the batch axis of topdown`` -- what the
  reference's MapBuilder.merge computes for maps that share one frame
  (element-wise max).  The fused map of each rank is what a
  multi-GPU job all-reduces (``parallel.all_reduce_fused``).  ``fused_out`` =
  (float32 (C,mh,mw), bool (C,mh,mw)) writes the fused map and its mask into
  caller-owned buffers, e.g. slots of a ring that is all-reduced once per several
  steps (fewer, larger collectives).  ``out`` = (float32 (b,C,mh,mw), bool (b,C,mh,mw)) does
  the same for the per-frame maps and masks (a caller that keeps a ring of output sets pays no
  allocation per call)."""
  return _orth_project(depth_map, value_map, valid_map, cam_pose, width_offset, height_offset,
                       cam_pitch, cam_height, map_res, map_width, map_height, focal_x, focal_y,
                       center_x, center_y, trunc_depth_min, trunc_depth_max, trunc_height_max,
                       clip_border, to_global, flip_h, fill_value, reduction, False, device,
                       True, fused_out, out)


def orth_project_and_flow(depth_map, trans_pose, valid_map, cam_pose, width_offset, height_offset,
                          cam_pitch, cam_height, map_res, map_width, map_height, focal_x, focal_y,
                          center_x, center_y, trunc_depth_min, trunc_depth_max, trunc_height_max,
                          clip_border, to_global=True, flip_h=True, fill_value=NINF, reduction=None,
                          device=None, out=None
                          ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
  """``orth_project`` (heights) and ``camera_affine_grid`` of the SAME depth maps in one call:
  ``(topdown (b,c,mh,mw), mask, grid (b,c,h,w,2))`` -- what the reference's ego-flow demo
  computes per frame with two calls (demos/ego_flow/run.py:75-90, maps.py:127-351 and 353-460).
  ``trans_pose`` = [x, z, yaw] is the camera's motion (``camera_affine_grid``'s argument), the
  other arguments are ``orth_project``'s.  One native call (dm_orth_project_flow_f32): the
  projection's launches and the flow kernel behind them; the library's LDS-window kernel for lean
  height maps can also compute the flow from the depth it has loaded (one depth read for both,
  ``dm_debug_flow_fused(1)``), which measured slower on MI355X than the two kernels and is off by
  default.  Bit-identical to the two separate calls either way.  ``out`` = (float32, bool) tensors of shape
  (b, c, mh, mw) to write the maps into."""
  call = _Call(depth_map, None, valid_map, cam_pose, width_offset, height_offset,
               cam_pitch, cam_height, map_res, map_width, map_height, focal_x, focal_y,
               center_x, center_y, trunc_depth_min, trunc_depth_max, trunc_height_max,
               clip_border, to_global, flip_h, fill_value, reduction, device)
  p = call.params
  B, dc, H, W = call.depth.shape
  shape = (p.B, call.oc, p.mh, p.mw)
  if out is not None:
    topdown, mask = out
    for t, dt in ((topdown, torch.float32), (mask, torch.bool)):
      if tuple(t.shape) != shape or t.dtype != dt or t.device != call.dev or not t.is_contiguous():
        raise ValueError(f"`out` must be contiguous {dt} {shape} tensors on {call.dev}")
  else:
    topdown = torch.empty(shape, dtype=torch.float32, device=call.dev)
    mask = torch.empty(shape, dtype=torch.bool, device=call.dev)
  flow_table = frames.build_frame_table(B, trans_pose, cam_pitch, cam_height, 0., 0., inverse_pitch=True)
  grid = torch.empty((B, dc, H, W, 2), dtype=torch.float32, device=call.dev)
  ws_bytes = max(call.ws_bytes, B * _native.FRAME_FLOATS * 4)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=call.dev)
  _native.check_status()
  with _on_device(call.dev):
    _native.check(_native.lib().dm_orth_project_flow_f32(
        ctypes.byref(p), _ptr(call.frames), _ptr(flow_table), _ptr(call.depth), _ptr(call.valid),
        _ptr(topdown), _ptr(mask), None, None, _ptr(grid), _ptr(ws), ws_bytes, _native.status_ptr(),
        _stream_ptr(call.dev)))
  if call.target != call.dev:
    topdown, mask, grid = topdown.to(call.target), mask.to(call.target), grid.to(call.target)
  return topdown, mask, grid


def orth_project_fused(
    depth_map, value_map, valid_map, cam_pose, width_offset, height_offset, cam_pitch,
    cam_height, map_res, map_width, map_height, focal_x, focal_y, center_x, center_y,
    trunc_depth_min, trunc_depth_max, trunc_height_max, clip_border, to_global=True,
    flip_h=True, fill_value=NINF, reduction=None, out: Optional[torch.Tensor] = None,
    device=None
) -> Tuple[torch.Tensor, torch.Tensor]:
  """Project a batch and fuse all frames into ONE (C, mh, mw) map (max/min).

  New entry point (north_star "projected+fused"): the per-rank partial global
  map.  Identical to reducing ``orth_project``'s output over the batch axis,
  which in turn equals the reference's MapBuilder.merge for maps that share
  resolution/offsets/size (element-wise max).  ``out`` (C, mh, mw),
  if given, is the running world map: its content takes part in the reduction.
  """
  import ctypes
  call = _Call(depth_map, value_map, valid_map, cam_pose, width_offset, height_offset,
               cam_pitch, cam_height, map_res, map_width, map_height, focal_x, focal_y,
               center_x, center_y, trunc_depth_min, trunc_depth_max, trunc_height_max,
               clip_border, to_global, flip_h, fill_value, reduction, device)
  p = call.params
  shape = (call.oc, p.mh, p.mw)
  accumulate = out is not None
  if accumulate:
    if tuple(out.shape) != shape or out.dtype != torch.float32 or out.device != call.dev \
        or not out.is_contiguous():
      raise ValueError(f"`out` must be a contiguous float32 {shape} tensor on {call.dev}")
  else:
    out = torch.empty(shape, dtype=torch.float32, device=call.dev)
  mask = torch.empty(shape, dtype=torch.bool, device=call.dev)
  ws, ws_bytes = call.workspace()
  _native.check_status()
  with _on_device(call.dev):
    _native.check(_native.lib().dm_orth_project_fused_f32(
        ctypes.byref(p), _ptr(call.frames), _ptr(call.depth), _ptr(call.value),
        _ptr(call.valid), _ptr(out), _ptr(mask), int(accumulate), _ptr(ws), ws_bytes,
        _native.status_ptr(), _stream_ptr(call.dev)))
  if call.target != call.dev:       # outputs live on the depth map's device (maps.py:227-232)
    out, mask = out.to(call.target), mask.to(call.target)
  return out, mask


class PreparedProjection:
  """The camera state of a batch validated once and kept in a small device buffer (48 bytes per
  frame, ``dm_frames_prepare_f32``), then projected any number of times: ``orth_project`` /
  ``orth_project_and_fuse`` on it enqueue the kernels and nothing else -- no host-side frame
  table, no validation -- so the calls are cheap on the host, and because the kernels read the
  poses from that buffer (not from their arguments, as plain calls do) a launch sequence captured
  into a HIP graph (``torch.cuda.graph``) projects whatever poses the buffer holds when it is
  replayed.  New poses for the same shapes: ``update(cam_pose=...)`` -- the launch plan is
  derived on the host first and the buffer is written only if the plan is unchanged.

  Made by ``MapProjector.prepare`` / ``prepare_orth_project``.  Applies where the library's
  strip path does (max / min, ``trunc_depth_min >= 0`` and ``trunc_depth_max`` given, map and
  image widths multiples of 4, one camera pitch for the batch); otherwise preparing raises
  ``NativeError`` and plain ``orth_project`` is the way.  One projection at a time per object
  (its workspace is its own).  Results are those of ``orth_project`` bit for bit.
  """

  def __init__(self, batch, height, width, depth_channels, value_channels, valid_channels, cam_pose,
               width_offset, height_offset, cam_pitch, cam_height, map_res, map_width, map_height,
               focal_x, focal_y, center_x, center_y, trunc_depth_min, trunc_depth_max,
               trunc_height_max, clip_border, to_global, flip_h, fill_value, reduction, device):
    import ctypes
    dev = _compute_device(torch.device(device) if device is not None else torch.device("cuda"))
    self.dev = dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
    self.shape_in = (int(batch), int(depth_channels), int(height), int(width))
    self.vc, self.valid_c = int(value_channels), int(valid_channels)
    key = (int(batch), int(depth_channels), self.vc, int(height), int(width), int(map_height),
           int(map_width), int(clip_border) if clip_border is not None else 0, bool(flip_h),
           bool(to_global), _reduction_code(reduction), trunc_depth_min, trunc_depth_max,
           trunc_height_max, self.valid_c, float(center_x), float(center_y), float(focal_x),
           float(focal_y), float(map_res), fill_value)
    self.params, self.ws_bytes = _make_params(key)
    self.oc = self.vc if self.vc else int(depth_channels)
    self.fill_value = fill_value
    self._camera = dict(cam_pitch=cam_pitch, cam_height=cam_height, width_offset=width_offset,
                        height_offset=height_offset)
    lib = _native.lib()
    nbytes = int(lib.dm_frames_prepared_bytes(ctypes.byref(self.params)))
    if nbytes == 0:
      raise _native.NativeError("these parameters cannot be prepared (the strip path does not "
                                "apply to them): use orth_project")
    with _on_device(self.dev):
      self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
      self.ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=self.dev)
    self.plan = None
    self._frames = None
    self.update(cam_pose)

  def update(self, cam_pose=None, **camera) -> "PreparedProjection":
    """Upload other poses (and optionally ``cam_pitch`` / ``cam_height`` / offsets) for the same
    shapes.  Raises -- before anything is written to the device -- if they need a different
    launch plan than the one in use (e.g. another pitch: make a new PreparedProjection)."""
    import ctypes
    camera = {k: v for k, v in camera.items() if v is not None}
    unknown = set(camera) - set(self._camera)
    if unknown:
      raise TypeError(f"update() got unexpected keyword arguments {sorted(unknown)}")
    merged = dict(self._camera, **camera)
    table = frames.build_frame_table(self.shape_in[0], cam_pose, merged["cam_pitch"],
                                     merged["cam_height"], merged["width_offset"],
                                     merged["height_offset"])
    plan = _native.FramesPlan()
    with _on_device(self.dev):
      rc = _native.lib().dm_frames_prepare_f32(
          ctypes.byref(self.params), _ptr(table), _ptr(self.buf), self.buf.numel(),
          None if self.plan is None else ctypes.byref(self.plan), ctypes.byref(plan),
          _stream_ptr(self.dev))
    _native.check(rc)       # (a mismatch leaves the buffer, the plan and the camera as they were)
    self._camera = merged
    self.plan, self._frames = plan, table        # (the host table must outlive the async copy)
    return self

  def status(self) -> int:
    """Bits a projection raised so far (0: none).  Synchronises the device first, so that every
    projection enqueued before the call has reported; the same bits make the NEXT projection
    call of the process raise ``NativeError`` without any synchronisation."""
    torch.cuda.synchronize(self.dev)
    return int(_native.status_word()[1][0])

  def _check(self, t, channels, dtype, what):
    B, _, H, W = self.shape_in
    if not (torch.is_tensor(t) and t.device == self.dev and t.dtype == dtype and t.is_contiguous()
            and tuple(t.shape) == (B, channels, H, W)):
      raise ValueError(f"{what} must be a contiguous {dtype} tensor of shape "
                       f"{(B, channels, H, W)} on {self.dev}")

  def _run(self, depth_map, value_map, valid_map, out, height, fused):
    import ctypes
    p = self.params
    self._check(depth_map, self.shape_in[1], torch.float32, "depth_map")
    if (value_map is not None) != bool(self.vc):
      raise ValueError(f"prepared for {self.vc} value channels")
    if value_map is not None:
      self._check(value_map, self.vc, torch.float32, "value_map")
    if (valid_map is not None) != bool(self.valid_c):
      raise ValueError(f"prepared for {self.valid_c} valid-map channels")
    if valid_map is not None:
      self._check(valid_map, self.valid_c, torch.bool, "valid_map")
    shape = (p.B, self.oc, p.mh, p.mw)
    if out is None:
      out = (torch.empty(shape, dtype=torch.float32, device=self.dev),
             torch.empty(shape, dtype=torch.bool, device=self.dev))
    top, mask = out
    _native.check_status()
    with _on_device(self.dev):
      _native.check(_native.lib().dm_orth_project_prepared_f32(
          ctypes.byref(p), ctypes.byref(self.plan), _ptr(self.buf), _ptr(depth_map), _ptr(value_map),
          _ptr(valid_map), _ptr(top), _ptr(mask), _ptr(height),
          None if fused is None else _ptr(fused[0]), None if fused is None else _ptr(fused[1]),
          _ptr(self.ws), self.ws.numel(), _native.status_ptr(), _stream_ptr(self.dev)))
    return top, mask

  def orth_project(self, depth_map, value_map=None, valid_map=None, get_height_map=False, out=None):
    """``orth_project`` of GPU tensors with the prepared camera state: ``(topdown, mask[,
    height_map])``.  ``out`` = (float32, bool) tensors of shape (b, C, mh, mw) to write into
    (static outputs, e.g. under graph capture)."""
    height = None
    if get_height_map and self.vc:
      height = torch.empty((self.params.B, self.shape_in[1], self.params.mh, self.params.mw),
                           dtype=torch.float32, device=self.dev)
    top, mask = self._run(depth_map, value_map, valid_map, out, height, None)
    if not get_height_map:
      return top, mask
    return top, mask, (top if height is None else torch.broadcast_to(height, top.shape))

  def orth_project_and_fuse(self, depth_map, value_map=None, valid_map=None, out=None, fused_out=None):
    """Per-frame maps plus the batch-fused map (max / min over the frames):
    ``(topdown, mask, fused, fused_mask)``."""
    p = self.params
    if fused_out is None:
      fused_out = (torch.empty((self.oc, p.mh, p.mw), dtype=torch.float32, device=self.dev),
                   torch.empty((self.oc, p.mh, p.mw), dtype=torch.bool, device=self.dev))
    top, mask = self._run(depth_map, value_map, valid_map, out, None, fused_out)
    return top, mask, fused_out[0], fused_out[1]


def prepare_orth_project(batch, height, width, cam_pose, width_offset, height_offset, cam_pitch,
                         cam_height, map_res, map_width, map_height, focal_x, focal_y, center_x,
                         center_y, trunc_depth_min, trunc_depth_max, trunc_height_max, clip_border,
                         to_global, flip_h=True, fill_value=None, reduction=None, device=None,
                         depth_channels=1, value_channels=0, valid_channels=0) -> PreparedProjection:
  """Upload the camera state of ``batch`` frames of ``height`` x ``width`` pixels once; see
  ``PreparedProjection``.  Arguments as ``orth_project``'s (reference maps.py:127-153)."""
  return PreparedProjection(batch, height, width, depth_channels, value_channels, valid_channels,
                            cam_pose, width_offset, height_offset, cam_pitch, cam_height, map_res,
                            map_width, map_height, focal_x, focal_y, center_x, center_y,
                            trunc_depth_min, trunc_depth_max, trunc_height_max, clip_border,
                            to_global, flip_h, fill_value, reduction, device)


def fuse_batch(maps: torch.Tensor, reduction=None, out: Optional[torch.Tensor] = None
               ) -> torch.Tensor:
  """Fuse maps (B, C, mh, mw) that share one frame into one (C, mh, mw) map:
  element-wise max (or min) over the batch axis on the GPU (dm_fuse_batch_f32).
  ``out``, if given, is a running world map whose content takes part."""
  if maps.device.type != "cuda" or maps.dtype != torch.float32:
    raise RuntimeError("fuse_batch expects a float32 GPU tensor")
  maps = maps.contiguous()
  shape = tuple(maps.shape[1:])
  accumulate = out is not None
  if accumulate:
    if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() \
        or out.device != maps.device:
      raise ValueError(f"`out` must be a contiguous float32 {shape} tensor on {maps.device}")
  else:
    out = torch.empty(shape, dtype=torch.float32, device=maps.device)
  n = int(np.prod(shape)) if shape else 1
  with _on_device(maps.device):
    _native.check(_native.lib().dm_fuse_batch_f32(
        _ptr(maps), maps.shape[0], n, _ptr(out), _reduction_code(reduction), int(accumulate),
        _stream_ptr(maps.device)))
  return out


def mask_from_map(topdown: torch.Tensor, fill_value: Optional[float]) -> torch.Tensor:
  """mask = (map - fill != 0, NaN -> False): reference utils.py:489-491 as a
  function of the finished map (used after a cross-rank max all-reduce)."""
  if topdown.device.type != "cuda":
    raise RuntimeError("mask_from_map runs on the GPU")
  t = topdown.contiguous()
  mask = torch.empty(t.shape, dtype=torch.bool, device=t.device)
  with _on_device(t.device):
    _native.check(_native.lib().dm_mask_from_map_f32(
        _ptr(t), 0.0 if fill_value is None else float(fill_value), _ptr(mask), t.numel(),
        _stream_ptr(t.device)))
  return mask


SAMPLE_MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2}


def crop_sample(image: torch.Tensor, center: torch.Tensor, crop_width: int, crop_height: int,
                fill_value: Optional[float] = None, mask: Optional[torch.Tensor] = None, mode: str = "nearest"):
  """generate_crop_grid + image_sample(mode=...) (reference utils.py:571-652) as one HIP kernel per image
  (dm_crop_sample_f32): 'nearest' is crop_nearest; 'bilinear' / 'bicubic' follow torch's
  grid_sample(align_corners=True) on the image padded by one pixel (float32, within a few ulp of torch's CPU
  kernel; NaN / inf where an empty cell meets a zero weight, as there).  The optional bool ``mask`` is sampled
  on the same coordinates with fill False and comes back as "nonzero".  Returns the crop (and the mask's)."""
  if mode not in SAMPLE_MODES:
    raise ValueError(f"mode must be one of {sorted(SAMPLE_MODES)}, not {mode!r}")
  if mode == "nearest":
    return crop_nearest(image, center, crop_width, crop_height, fill_value=fill_value, mask=mask)
  if image.device.type != "cuda":
    raise RuntimeError("crop_sample runs on the GPU")
  img = image.to(torch.float32).contiguous()
  b, c, h, w = img.shape
  ctr = utils.to_tensor(center, device=img.device).to(torch.float32).reshape(-1, 2)
  if ctr.shape[0] == 1 and b > 1:
    ctr = ctr.expand(b, 2)
  assert ctr.shape[0] == b, (ctr.shape, b)
  ctr = ctr.contiguous()
  out = torch.empty((b, c, crop_height, crop_width), dtype=torch.float32, device=img.device)
  msk = out_mask = None
  if mask is not None:
    msk = mask.to(device=img.device, dtype=torch.bool).expand(img.shape).contiguous()
    out_mask = torch.empty(out.shape, dtype=torch.bool, device=img.device)
  with _on_device(img.device):
    _native.check(_native.lib().dm_crop_sample_f32(
        _ptr(img), _ptr(msk), _ptr(ctr), b, c, h, w, crop_height, crop_width,
        0.0 if fill_value is None else float(fill_value), 0 if fill_value is None else 1, SAMPLE_MODES[mode],
        _ptr(out), _ptr(out_mask), _stream_ptr(img.device)))
  return (out, out_mask) if mask is not None else out


def crop_nearest(image: torch.Tensor, center: torch.Tensor, crop_width: int, crop_height: int,
                 fill_value: Optional[float] = None,
                 mask: Optional[torch.Tensor] = None):
  """generate_crop_grid + image_sample(mode='nearest') (reference utils.py:571-652) as one
  HIP gather: ``image`` (b, c, h, w) float32 on the GPU, ``center`` (b, 2) crop centres in
  pixels, optional bool ``mask`` of the same shape sampled on the same coordinates with
  fill False.  Returns the crop (and the cropped mask)."""
  if image.device.type != "cuda":
    raise RuntimeError("crop_nearest runs on the GPU")
  img = image.to(torch.float32).contiguous()
  b, c, h, w = img.shape
  ctr = utils.to_tensor(center, device=img.device).to(torch.float32).reshape(-1, 2)
  if ctr.shape[0] == 1 and b > 1:
    ctr = ctr.expand(b, 2)
  assert ctr.shape[0] == b, (ctr.shape, b)
  ctr = ctr.contiguous()
  out = torch.empty((b, c, crop_height, crop_width), dtype=torch.float32, device=img.device)
  msk = out_mask = None
  if mask is not None:
    msk = mask.to(device=img.device, dtype=torch.bool).expand(img.shape).contiguous()
    out_mask = torch.empty(out.shape, dtype=torch.bool, device=img.device)
  with _on_device(img.device):
    _native.check(_native.lib().dm_crop_nearest_f32(
        _ptr(img), _ptr(msk), _ptr(ctr), b, c, h, w, crop_height, crop_width,
        0.0 if fill_value is None else float(fill_value), 0 if fill_value is None else 1,
        _ptr(out), _ptr(out_mask), _stream_ptr(img.device)))
  return (out, out_mask) if mask is not None else out


# ---------------------------------------------------------------------------
# Point-set helpers (small tensors; float32 formulas in the reference's order)
# ---------------------------------------------------------------------------
def _points(points, device=None) -> torch.Tensor:
  t = utils.to_tensor(points, device=device).to(torch.float32)
  return t.reshape(1, 3) if t.dim() < 2 else t


def image_to_camera_space(points, focal_x, focal_y, center_x, center_y, flip_h=True,
                          height=None, device=None, _validate_args=True) -> torch.Tensor:
  """(col, row, depth) -> camera space X right / Y up / Z forward
  (reference maps.py:616-682)."""
  pts = utils.to_tensor(points, device=device).to(torch.float32)
  shape = pts.shape
  if pts.dim() < 2:
    pts = pts.reshape(-1, 3)
  if flip_h and height is None:
    if pts.dim() < 3:
      raise RuntimeError("The rank of `points` must be at least 3D (..., h, w, 3) "
                         "or `height` should be provided if `flip_h` is enabled.")
    height = pts.shape[-3]
  x, y, z = pts.unbind(-1)
  if flip_h:
    y = (height - 1) - y
  cx, cy, fx, fy = (utils.to_tensor_like(v, pts) for v in (center_x, center_y, focal_x, focal_y))
  out = torch.stack(((x - cx) / fx * z, (y - cy) / fy * z, z), dim=-1)
  return out.reshape(shape)


def camera_to_image_space(points, focal_x, focal_y, center_x, center_y, flip_h=True,
                          height=None, device=None, _validate_args=True) -> torch.Tensor:
  """Camera space -> (col, row, depth) (reference maps.py:684-751)."""
  pts = utils.to_tensor(points, device=device).to(torch.float32)
  shape = pts.shape
  if pts.dim() < 2:
    pts = pts.reshape(-1, 3)
  if flip_h and height is None:
    if pts.dim() < 3:
      raise RuntimeError("The rank of `points` must be at least 3D (..., h, w, 3) "
                         "or `height` should be provided if `flip_h` is enabled.")
    height = pts.shape[-3]
  x, y, z = pts.unbind(-1)
  cx, cy, fx, fy = (utils.to_tensor_like(v, pts) for v in (center_x, center_y, focal_x, focal_y))
  z_eps = z + 1e-7
  u = x / z_eps * fx + cx
  v = y / z_eps * fy + cy
  if flip_h:
    v = (height - 1) - v
  return torch.stack((u, v, z), dim=-1).reshape(shape)


def _cpu_f32(v, width: Optional[int] = None) -> torch.Tensor:
  t = utils.to_tensor(v).detach().to(device="cpu", dtype=torch.float32)
  return t.reshape(-1) if width is None else t.reshape(-1, width)


def _rows(t: torch.Tensor, batch: int) -> torch.Tensor:
  if t.shape[0] == 1 and batch != 1:
    t = t.expand(batch, *t.shape[1:])
  if t.shape[0] != batch:
    raise ValueError(f"per-batch argument has {t.shape[0]} rows, expected 1 or {batch}")
  return t


def _affine(points: torch.Tensor, axis, angle, offset: torch.Tensor,
            translate_first: bool) -> torch.Tensor:
  """rotate(points) + offset, or rotate(points + offset): reference
  utils.rotate / utils.translate (utils.py:229-330) with the rotation matrix
  built on the CPU.  GPU tensors go through dm_affine_points_f32 (the exact FMA
  chain of the reference's CPU bmm); CPU tensors use torch (= the reference)."""
  shape = points.shape
  b = shape[0]
  angle_cpu = _cpu_f32(angle)
  offset_cpu = _rows(_cpu_f32(offset, 3), b)
  if points.device.type != "cuda":
    flat = points.reshape(b, -1, 3)
    if translate_first:
      flat = utils.translate(flat, offset_cpu)
    flat = utils.rotate(flat, axis, angle_cpu)
    if not translate_first:
      flat = utils.translate(flat, offset_cpu)
    return flat.reshape(shape)
  rot = _rows(utils.rotation_matrix(torch.tensor([axis]), angle_cpu).reshape(-1, 9), b)
  dev = points.device
  pts = points.reshape(b, -1, 3).contiguous()
  out = torch.empty_like(pts)
  rot_d = rot.contiguous().to(dev)
  off_d = offset_cpu.contiguous().to(dev)
  with _on_device(dev):
    _native.check(_native.lib().dm_affine_points_f32(
        _ptr(pts), _ptr(rot_d), _ptr(off_d), b, pts.shape[1], int(translate_first), _ptr(out),
        _stream_ptr(dev)))
  return out.reshape(shape)


def _lift(cam_height, sign: float) -> torch.Tensor:
  h = _cpu_f32(cam_height)
  o = torch.zeros_like(h)
  return torch.stack((o, sign * h, o), dim=-1)


def camera_to_local_space(points, cam_pitch, cam_height, device=None,
                          _validate_args=True) -> torch.Tensor:
  """Rotate by the camera pitch about X, then lift by the camera height
  (reference maps.py:753-800)."""
  pts = _points(points, device)
  return _affine(pts, [1., 0., 0.], _cpu_f32(cam_pitch), _lift(cam_height, 1.0), False)


def local_to_camera_space(points, cam_pitch, cam_height, device=None,
                          _validate_args=True) -> torch.Tensor:
  """Drop by the camera height, then rotate by -pitch (reference maps.py:802-848)."""
  pts = _points(points, device)
  return _affine(pts, [1., 0., 0.], -_cpu_f32(cam_pitch), _lift(cam_height, -1.0), True)


def _pose_offset(pose: torch.Tensor) -> torch.Tensor:
  return torch.stack((pose[:, 0], torch.zeros_like(pose[:, 0]), pose[:, 1]), dim=-1)


def local_to_global_space(points, cam_pose, device=None, _validate_args=True) -> torch.Tensor:
  """Rotate by yaw about Y, then translate by (pose_x, 0, pose_z)
  (reference maps.py:850-895)."""
  pts = _points(points, device)
  pose = _cpu_f32(cam_pose, 3)
  return _affine(pts, [0., 1., 0.], pose[:, 2], _pose_offset(pose), False)


def global_to_local_space(points, cam_pose, device=None, _validate_args=True) -> torch.Tensor:
  """Translate by -(pose_x, 0, pose_z), then rotate by -yaw (reference maps.py:897-942)."""
  pts = _points(points, device)
  pose = _cpu_f32(cam_pose, 3)
  return _affine(pts, [0., 1., 0.], -pose[:, 2], -_pose_offset(pose), True)


def _offset_column(v, like: torch.Tensor) -> torch.Tensor:
  t = utils.to_tensor(v, device=like.device).to(torch.float32).reshape(-1)
  return t.reshape((-1,) + (1,) * (like.dim() - 1))


def map_quantize(x_coords, z_coords, width_offset, height_offset, map_res,
                 map_height=None, flip_h=True, device=None, _validate_args=True
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
  """World x/z -> integer map column/row, round-half-up
  (reference maps.py:944-1019)."""
  x = utils.to_tensor(x_coords, device=device).to(torch.float32)
  z = utils.to_tensor(z_coords, device=x.device).to(torch.float32)
  x, z = torch.broadcast_tensors(x, z)
  if x.dim() < 2:
    x, z = x.reshape(1, -1), z.reshape(1, -1)
  if x.device.type == "cuda":
    return _map_quantize_native(x, z, width_offset, height_offset, map_res, map_height, flip_h)
  col = x / map_res + _offset_column(width_offset, x)
  row = z / map_res + _offset_column(height_offset, x)
  if flip_h:
    assert map_height is not None
    row = (torch.tensor(map_height, device=x.device) - 1) - row
  return (torch.floor(col + 0.5).to(torch.int64), torch.floor(row + 0.5).to(torch.int64))


def _map_quantize_native(x, z, width_offset, height_offset, map_res, map_height, flip_h):
  """GPU tensors: dm_map_quantize_f32 (true IEEE division -- PyTorch's own GPU
  division by a scalar multiplies by the reciprocal and can flip cells)."""
  b = x.shape[0]
  dev = x.device
  xs = x.reshape(b, -1).contiguous()
  zs = z.reshape(b, -1).contiguous()
  woff = _rows(_cpu_f32(width_offset), b).contiguous().to(dev)
  hoff = _rows(_cpu_f32(height_offset), b).contiguous().to(dev)
  if flip_h:
    assert map_height is not None
  xb = torch.empty(xs.shape, dtype=torch.int64, device=dev)
  zb = torch.empty(xs.shape, dtype=torch.int64, device=dev)
  with _on_device(dev):
    _native.check(_native.lib().dm_map_quantize_f32(
        _ptr(xs), _ptr(zs), _ptr(woff), _ptr(hoff), b, xs.shape[1], float(map_res),
        int(map_height) if map_height is not None else 1, int(bool(flip_h)), _ptr(xb), _ptr(zb),
        _stream_ptr(dev)))
  return xb.reshape(x.shape), zb.reshape(x.shape)


def map_dequantize(x_coords, z_coords, width_offset, height_offset, map_res,
                   map_height=None, flip_h=True, device=None, _validate_args=True
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
  """Inverse of map_quantize (cell -> world x/z of its centre)."""
  col = utils.to_tensor(x_coords, device=device).to(torch.float32)
  row = utils.to_tensor(z_coords, device=col.device).to(torch.float32)
  col, row = torch.broadcast_tensors(col, row)
  if col.dim() < 2:
    col, row = col.reshape(1, -1), row.reshape(1, -1)
  if flip_h:
    assert map_height is not None
    row = (torch.tensor(map_height, device=col.device) - 1) - row
  z = (row - _offset_column(height_offset, col)) * map_res
  x = (col - _offset_column(width_offset, col)) * map_res
  return x, z


def depth_map_to_point_cloud(depth_map, valid_map, focal_x, focal_y, center_x, center_y,
                             trunc_depth_min, trunc_depth_max, flip_h=True, device=None,
                             _validate_args=True) -> Tuple[torch.Tensor, torch.Tensor]:
  """Camera-space point cloud (..., h, w, 3) + validity (..., h, w) of a depth
  map (reference maps.py:462-545).  Materialises the cloud -- orth_project does
  NOT go through this; it is here for callers that want the points."""
  depth = utils.to_4D_image(utils.to_tensor(depth_map, device=device)).to(torch.float32)
  xs, ys = utils.generate_image_coords(depth.shape, torch.float32, depth.device)
  cloud = image_to_camera_space(torch.stack((xs, ys, depth), dim=-1), focal_x, focal_y,
                                center_x, center_y, flip_h, height=depth.shape[-2])
  ok = torch.ones_like(depth, dtype=torch.bool)
  if trunc_depth_max is not None:
    ok = ok & (depth <= trunc_depth_max)
  if trunc_depth_min is not None:
    ok = ok & (depth >= trunc_depth_min)
  if valid_map is not None:
    vm = utils.to_4D_image(utils.to_tensor(valid_map, device=depth.device)).to(torch.bool)
    ok = ok & vm
  return cloud, ok


def height_map_to_point_cloud(height_map, width_offset, height_offset, map_res, map_height,
                              flip_h=True, device=None, _validate_args=True) -> torch.Tensor:
  """Cell-centre point cloud (b, c, h, w, 3) of a height map
  (reference maps.py:547-612)."""
  hm = utils.to_4D_image(utils.to_tensor(height_map, device=device)).to(torch.float32)
  cols, rows = utils.generate_image_coords(hm.shape, torch.float32, hm.device)
  x, z = map_dequantize(cols, rows, width_offset, height_offset, map_res, map_height, flip_h)
  return torch.stack((x, hm, z), dim=-1)


def compute_center_offsets(cam_pose, width_offset, height_offset, map_res, map_width,
                           map_height, to_global, center_mode=CenterMode.none, device=None,
                           _validate_args=True) -> Tuple[torch.Tensor, torch.Tensor]:
  """Offsets that put the global origin / the camera at the map centre
  (reference maps.py:1175-1248)."""
  mode = center_mode if type(center_mode) is CenterMode else CenterMode(center_mode)
  if mode is CenterMode.none and device is None and (not torch.is_tensor(cam_pose) or cam_pose.device.type == "cpu") \
      and isinstance(width_offset, (int, float)) and isinstance(height_offset, (int, float)):
    # (the demo's call -- numbers in, nothing to centre: the two 0-d float32 tensors the general route returns)
    return torch.tensor(width_offset, dtype=torch.float32), torch.tensor(height_offset, dtype=torch.float32)
  pose = utils.to_tensor(torch.zeros(3) if cam_pose is None else cam_pose,
                         device=device).to(torch.float32)
  woff = utils.to_tensor(0. if width_offset is None else width_offset,
                         device=pose.device).to(torch.float32)
  hoff = utils.to_tensor(0. if height_offset is None else height_offset,
                         device=pose.device).to(torch.float32)
  if mode is CenterMode.none:
    return woff + 0., hoff + 0.
  centre = torch.zeros_like(pose)
  if mode is CenterMode.camera and to_global:
    centre = local_to_global_space(centre, pose)
  col, row = map_quantize(centre[..., 0], centre[..., 2], 0., 0., map_res, map_height,
                          flip_h=False)
  return woff + (map_width / 2. - col), hoff + (map_height / 2. - row)


# ---------------------------------------------------------------------------
# project / scatter (reference maps.py:1089-1173, utils.py:389-492)
# ---------------------------------------------------------------------------
def _scatter_flat(canvas: torch.Tensor, flat_index: torch.Tensor, values: torch.Tensor,
                  n_data: int, fill_value, reduction) -> Tuple[torch.Tensor, torch.Tensor]:
  """canvas (b..., d1..dn) with n_data trailing data dims, flat_index (b..., N)
  int64 (< 0 = dropped), values (b..., N).  Runs dm_scatter_f32 on the GPU."""
  import ctypes
  target = canvas.device
  dev = _compute_device(target)
  batch_shape = tuple(canvas.shape[:canvas.dim() - n_data])
  M = int(np.prod(canvas.shape[canvas.dim() - n_data:]))
  N = values.shape[-1]
  vshape = torch.broadcast_shapes(tuple(values.shape[:-1]), batch_shape)
  if tuple(vshape) != batch_shape:
    raise ValueError(f"values {tuple(values.shape)} do not broadcast to canvas batch {batch_shape}")
  C = batch_shape[-1] if len(batch_shape) >= 1 else 1
  R = int(np.prod(batch_shape[:-1])) if len(batch_shape) >= 1 else 1
  vals = values.to(device=dev, dtype=torch.float32).expand(*batch_shape, N).contiguous()
  idx = flat_index.to(device=dev, dtype=torch.int64)
  # index rows either follow every channel or are shared by all channels of a row
  if idx.dim() == len(batch_shape) + 1 and len(batch_shape) >= 1 and idx.shape[-2] == 1 and C > 1:
    idx = idx.expand(*batch_shape[:-1], 1, N).contiguous()
    Ci = 1
  else:
    idx = idx.expand(*batch_shape, N).contiguous()
    Ci = C
  out = canvas.to(device=dev, dtype=torch.float32).contiguous().clone()
  mask = torch.empty(out.shape, dtype=torch.bool, device=dev)
  has_fill = fill_value is not None
  red = _reduction_code(reduction)
  lib = _native.lib()
  need = lib.dm_scatter_workspace_bytes(R, C, M, int(has_fill), red)
  ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
  with _on_device(dev):
    _native.check(lib.dm_scatter_f32(
        _ptr(vals), _ptr(idx), _ptr(out), _ptr(mask), R, C, Ci, N, M,
        float(fill_value) if has_fill else 0.0, int(has_fill), red, _ptr(ws), need,
        _stream_ptr(dev)))
  if target != dev:
    out, mask = out.to(target), mask.to(target)
  return out, mask


def scatter_nd(canvas, indices, values, masks=None, fill_value=None, reduction=None):
  """utils.scatter_tensor (reference utils.py:389-492): canvas (b..., d1..dn),
  indices (b..., N, n), values (b..., N), masks (b..., N)."""
  canvas = utils.to_tensor(canvas)
  indices = utils.to_tensor(indices, device=canvas.device).to(torch.int64)
  values = utils.to_tensor(values, device=canvas.device)
  n = indices.shape[-1]
  if canvas.dim() <= n:
    raise AssertionError(f"The rank of `canvas` must be greater than {n}, got {canvas.dim()}")
  dims = canvas.shape[-n:]
  ok = torch.ones(indices.shape[:-1], dtype=torch.bool, device=canvas.device)
  if masks is not None:
    ok = ok & utils.to_tensor(masks, device=canvas.device).to(torch.bool)
  flat = torch.zeros(indices.shape[:-1], dtype=torch.int64, device=canvas.device)
  for i in range(n):
    di = indices[..., i]
    ok = ok & (di >= 0) & (di < dims[i])
    flat = flat * dims[i] + di
  flat = torch.where(ok, flat, torch.full_like(flat, -1))
  return _scatter_flat(canvas, flat, values, n, fill_value, reduction)


def project(coords, values, masks, canvas, canvas_masks=None, fill_value=None,
            reduction=None, device=None, _validate_args=True):
  """Scatter ``values`` (b, ..., n) onto ``canvas`` (b, ..., mh, mw) at
  ``coords`` (b, ..., n, 2) = (row, col) with a reduction; returns (canvas,
  mask).  Reference maps.py:1089-1173; runs on dm_scatter_f32."""
  coords = utils.to_tensor(coords, device=device)
  if coords.dim() < 3:
    coords = coords.reshape(1, -1, 2)
  dev = coords.device
  values = utils.to_tensor(values, device=dev).to(torch.float32)
  masks = utils.to_tensor(masks, device=dev).to(torch.bool)
  canvas = utils.to_tensor(canvas, device=dev).to(torch.float32)
  batch = torch.broadcast_shapes(tuple(values.shape), tuple(masks.shape),
                                 tuple(coords.shape[:-1]), tuple(canvas.shape[:-2]) + (1,))
  canvas = canvas.expand(*batch[:-1], *canvas.shape[-2:])
  out, changed = scatter_nd(canvas, coords.to(torch.int64), values, masks, fill_value, reduction)
  if canvas_masks is not None:
    cm = utils.to_tensor(canvas_masks, device=out.device).to(torch.bool)
    changed = torch.logical_or(torch.broadcast_to(cm, changed.shape), changed)
  return out, changed


def camera_affine_grid(depth_map, trans_pose, cam_pitch, cam_height, focal_x, focal_y,
                       center_x, center_y, flip_h=True, device=None, _validate_args=True):
  """Pixel coordinates (b, c, h, w, 2) every pixel of ``depth_map`` maps to
  after the camera moved by ``trans_pose`` = [x, z, yaw] (ego-motion flow field).
  Reference maps.py:353-460; one fused HIP kernel (dm_camera_affine_grid_f32)."""
  import ctypes
  first = depth_map if torch.is_tensor(depth_map) else utils.to_tensor(depth_map)
  target = torch.device(device) if device is not None else first.device
  dev = _compute_device(target)
  depth = _image(first, dev, torch.float32)
  B, dc, H, W = depth.shape
  p = _native.Params()
  p.B, p.dc, p.vc, p.H, p.W, p.mh, p.mw = B, dc, 0, H, W, 1, 1
  p.flip_h = int(bool(flip_h))
  p.cx, p.cy, p.fx, p.fy = float(center_x), float(center_y), float(focal_x), float(focal_y)
  p.res = 1.0
  table = frames.build_frame_table(B, trans_pose, cam_pitch, cam_height, 0., 0.,
                                   inverse_pitch=True)
  grid = torch.empty((B, dc, H, W, 2), dtype=torch.float32, device=dev)
  ws_bytes = B * _native.FRAME_FLOATS * 4
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
  with _on_device(dev):
    _native.check(_native.lib().dm_camera_affine_grid_f32(
        ctypes.byref(p), _ptr(table), _ptr(depth), _ptr(grid), _ptr(ws), ws_bytes,
        _stream_ptr(dev)))
  return grid if target == dev else grid.to(target)
