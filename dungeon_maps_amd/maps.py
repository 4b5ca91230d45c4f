"""Object API: MapProjector, TopdownMap, MapBuilder (+ crop / fuse).

Drop-in for reference ``dungeon_maps/maps.py:1252-2550``: same class names,
constructor keywords, properties and method names.  ``MapProjector`` is the
configuration holder: every functional API is reachable as a method whose
``None`` arguments fall back to the projector's stored defaults.
"""
import ctypes
import inspect
import threading
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import frames
from . import functional as F
from . import utils
from .functional import *  # noqa: F401,F403  (module-level functional twins)
from .functional import CenterMode, get
from .utils import NINF, CameraIntrinsics, Reduction

Float3D = Tuple[float, float, float]


class MapProjector:
  """Camera + map configuration and default-argument forwarding.

  Keyword names, defaults and meaning as reference maps.py:1253-1340
  (``fill_value`` defaults to NINF, ``flip_h`` to True, ``to_global`` to False).

  Performance note: the fast (LDS-windowed) kernels bound each image part's reach in the map
  through the depth range, so give ``trunc_depth_min`` (>= 0) and ``trunc_depth_max`` as the
  reference's demos do (0.15 / 5.05); without them large maps take the ~9x slower generic
  path (results are the same either way).
  """

  _FIELDS = ("width", "height", "hfov", "vfov", "cam_pose", "width_offset",
             "height_offset", "cam_pitch", "cam_height", "map_res", "map_width",
             "map_height", "trunc_depth_min", "trunc_depth_max", "trunc_height_max",
             "clip_border", "to_global", "flip_h", "fill_value", "reduction", "device")

  def __init__(self, width: int, height: int, hfov: float, vfov: Optional[float] = None,
               cam_pose: Optional[Float3D] = None, width_offset: Optional[float] = None,
               height_offset: Optional[float] = None, cam_pitch: Optional[float] = None,
               cam_height: Optional[float] = None, map_res: Optional[float] = None,
               map_width: Optional[int] = None, map_height: Optional[int] = None,
               trunc_depth_min: Optional[float] = None,
               trunc_depth_max: Optional[float] = None,
               trunc_height_max: Optional[float] = None,
               clip_border: Optional[int] = None, to_global: bool = False,
               flip_h: bool = True, fill_value: Optional[float] = NINF,
               reduction: Optional[Reduction] = None,
               device: Optional[torch.device] = None):
    values = locals()
    d = self.__dict__                      # (not through __setattr__: nothing is cached yet)
    for name in self._FIELDS:
      d[name] = values[name]
    d["_dm_defaults"] = {}                 # forwarding methods: their resolved defaults
    d["cam_params"] = utils.get_camera_intrinsics(width=width, height=height, hfov=hfov, vfov=vfov)
    d["_cam_key"] = (width, height, hfov, vfov)      # what cam_params was built from (clone)

  def __setattr__(self, name, value):
    # the forwarding methods cache their resolved defaults on the instance: any assignment
    # drops them
    d = self.__dict__
    cache = d.get("_dm_defaults")
    if cache:
      cache.clear()
    d[name] = value

  def clone(self, width: Optional[int] = None, height: Optional[int] = None,
            hfov: Optional[float] = None, vfov: Optional[float] = None,
            cam_pose: Optional[Float3D] = None, width_offset: Optional[float] = None,
            height_offset: Optional[float] = None, cam_pitch: Optional[float] = None,
            cam_height: Optional[float] = None, map_res: Optional[float] = None,
            map_width: Optional[int] = None, map_height: Optional[int] = None,
            trunc_depth_min: Optional[float] = None, trunc_depth_max: Optional[float] = None,
            trunc_height_max: Optional[float] = None, clip_border: Optional[int] = None,
            to_global: Optional[bool] = None, flip_h: Optional[bool] = None,
            fill_value: Optional[float] = None, reduction: Optional[Reduction] = None,
            device: Optional[torch.device] = None) -> "MapProjector":
    """Shallow copy with some fields replaced (None = keep); parameter list of reference
    maps.py:1349-1372."""
    overrides = locals()
    # (a copy of the fields with the given ones replaced, not a trip through __init__: MapBuilder
    # clones three projectors per frame)
    new = object.__new__(MapProjector)
    d = new.__dict__
    mine = self.__dict__
    for name in self._FIELDS:
      v = overrides[name]
      d[name] = mine[name] if v is None else v
    d["_dm_defaults"] = {}
    # the intrinsics are rebuilt as the reference's clone rebuilds them (maps.py:1349-1404 goes
    # through __init__) unless the four fields are the ones the cached cam_params came from --
    # a projector whose width / height / hfov / vfov were reassigned must not hand on stale ones
    key = (d["width"], d["height"], d["hfov"], d["vfov"])
    if key == mine.get("_cam_key"):
      d["cam_params"] = mine["cam_params"]         # (never modified in place)
    else:
      d["cam_params"] = utils.get_camera_intrinsics(width=d["width"], height=d["height"],
                                                    hfov=d["hfov"], vfov=d["vfov"])
    d["_cam_key"] = key
    return new


  def prepare(self, batch: int, cam_pose=None, value_channels: int = 0, valid_channels: int = 0,
              depth_channels: int = 1, **overrides) -> "F.PreparedProjection":
    """Upload the camera state of ``batch`` frames once and get an object whose
    ``orth_project`` / ``orth_project_and_fuse`` only enqueue kernels (cheap on the host, HIP
    graph capturable): see ``PreparedProjection``.  ``overrides``: any ``orth_project`` keyword
    (``cam_pitch``, ``map_res``, ``fill_value``, ...), default = this projector's field."""
    cam = self.cam_params
    intr = {"focal_x": cam.fx, "focal_y": cam.fy, "center_x": cam.cx, "center_y": cam.cy}
    names = ("width_offset", "height_offset", "cam_pitch", "cam_height", "map_res", "map_width",
             "map_height", "focal_x", "focal_y", "center_x", "center_y", "trunc_depth_min",
             "trunc_depth_max", "trunc_height_max", "clip_border", "to_global", "flip_h",
             "fill_value", "reduction", "device")
    unknown = set(overrides) - set(names) - {"height", "width"}
    if unknown:
      raise TypeError(f"prepare() got unexpected keyword arguments {sorted(unknown)}")
    kw = {n: get(overrides.get(n), intr[n] if n in intr else getattr(self, n)) for n in names}
    return F.prepare_orth_project(batch, get(overrides.get("height"), self.height),
                                  get(overrides.get("width"), self.width),
                                  get(cam_pose, self.cam_pose), depth_channels=depth_channels,
                                  value_channels=value_channels, valid_channels=valid_channels, **kw)


def _forwarding_method(fn, doc_ref: str):
  params = inspect.signature(fn).parameters
  names = tuple(params)
  index = {n: i for i, n in enumerate(names)}
  fn_defaults = tuple(None if p.default is inspect.Parameter.empty else p.default
                      for p in params.values())
  intr = {"focal_x": "fx", "focal_y": "fy", "center_x": "cx", "center_y": "cy"}
  # where each argument's fallback lives: 1 = projector field, 2 = intrinsic, 0 = none
  kinds = tuple(2 if n in intr else (1 if n in MapProjector._FIELDS else 0) for n in names)
  attrs = tuple(intr.get(n, n) for n in names)
  nargs = len(names)

  cache_key = fn.__name__

  def resolve(self):
    """(value for an argument that was not passed, value for one passed as None), per argument:
    the projector's field / intrinsic, and for arguments not passed at all the functional
    API's own default where the projector has none."""
    cam = self.cam_params
    field = []
    for i in range(nargs):
      kind = kinds[i]
      field.append(getattr(self, attrs[i]) if kind == 1 else
                   getattr(cam, attrs[i]) if kind == 2 else None)   # get(arg, self.<arg>)
    absent = tuple(fn_defaults[i] if v is None else v for i, v in enumerate(field))
    # (tagged with the owner: a copy.copy() of a projector shares this dict until it is cleared)
    cached = (absent, tuple(field), id(self))
    self.__dict__.setdefault("_dm_defaults", {})[cache_key] = cached   # dropped by __setattr__
    return cached

  def method(self, *args, **kwargs):
    given = len(args)
    if given > nargs:
      raise TypeError(f"{fn.__name__}() takes at most {nargs} arguments")
    cached = self.__dict__["_dm_defaults"].get(cache_key)
    if cached is None or cached[2] != id(self):
      cached = resolve(self)
    absent, field = cached[0], cached[1]
    call = list(absent)
    for i in range(given):
      v = args[i]
      call[i] = field[i] if v is None else v
    for k, v in kwargs.items():
      i = index.get(k)
      if i is None:
        raise TypeError(f"{fn.__name__}() got an unexpected keyword argument '{k}'")
      if i < given:
        raise TypeError(f"{fn.__name__}() got multiple values for argument '{k}'")
      call[i] = field[i] if v is None else v
    return fn(*call)

  # The signature the reference's method of the same name declares (maps.py:1406-1749): `self`,
  # then the functional twin's parameters, every one the projector can supply (or that the
  # reference defaults) optional = None.
  sig_params = [inspect.Parameter("self", inspect.Parameter.POSITIONAL_OR_KEYWORD)]
  for i, (n, p) in enumerate(params.items()):
    if kinds[i] or n in ("value_map", "valid_map"):
      default = None
    elif p.default is inspect.Parameter.empty:
      default = inspect.Parameter.empty
    else:
      default = p.default
    sig_params.append(inspect.Parameter(n, inspect.Parameter.POSITIONAL_OR_KEYWORD,
                                        default=default, annotation=p.annotation))
  method.__signature__ = inspect.Signature(
      sig_params, return_annotation=inspect.signature(fn).return_annotation)
  method.__name__ = fn.__name__
  method.__qualname__ = f"MapProjector.{fn.__name__}"
  method.__doc__ = (f"``{fn.__name__}`` with None arguments taken from this projector "
                    f"(reference {doc_ref}).\n\n" + (fn.__doc__ or ""))
  return method


for _fn, _ref in (
    (F.orth_project, "maps.py:1406-1465"),
    (F.orth_project_and_fuse, "new: per-frame maps + batch-fused map in one call"),
    (F.orth_project_fused, "new: batch-fused partial global map"),
    (F.orth_project_and_flow, "new: orth_project + camera_affine_grid of the same depth maps (maps.py:1406-1493)"),
    (F.camera_affine_grid, "maps.py:1467-1493"),
    (F.depth_map_to_point_cloud, "maps.py:1495-1521"),
    (F.height_map_to_point_cloud, "maps.py:1523-1543"),
    (F.image_to_camera_space, "maps.py:1545-1567"),
    (F.camera_to_image_space, "maps.py:1569-1591"),
    (F.camera_to_local_space, "maps.py:1593-1607"),
    (F.local_to_camera_space, "maps.py:1609-1623"),
    (F.local_to_global_space, "maps.py:1625-1637"),
    (F.global_to_local_space, "maps.py:1639-1651"),
    (F.map_quantize, "maps.py:1653-1675"),
    (F.map_dequantize, "maps.py:1677-1699"),
    (F.project, "maps.py:1701-1723"),
    (F.compute_center_offsets, "maps.py:1725-1749"),
):
  setattr(MapProjector, _fn.__name__, _forwarding_method(_fn, _ref))


class TopdownMap:
  """A top-down map with its mask, height map and the projector that made it
  (reference maps.py:1753-1955)."""

  def __init__(self, topdown_map: Optional[torch.Tensor] = None,
               mask: Optional[torch.Tensor] = None,
               height_map: Optional[torch.Tensor] = None,
               map_projector: Optional[MapProjector] = None,
               is_height_map: Optional[bool] = None):
    self._proj = map_projector
    self._topdown_map = topdown_map
    self._mask = mask
    self._height_map = height_map
    if is_height_map is None:
      is_height_map = topdown_map is not None and topdown_map is height_map
    self._is_height_map = is_height_map

  is_empty = property(lambda self: self._topdown_map is None)
  is_height_map = property(lambda self: self._is_height_map)
  map = property(lambda self: self._topdown_map)
  topdown_map = property(lambda self: self._topdown_map)
  mask = property(lambda self: self._mask)
  proj = property(lambda self: self._proj)

  @property
  def height_map(self) -> torch.Tensor:
    return self._topdown_map if self._is_height_map else self._height_map

  def get_camera(self) -> torch.Tensor:
    """Cell (col, row) of the camera, (b, 2) int64."""
    return self.get_coords(torch.zeros(3), is_global=False).squeeze(-2)

  def get_origin(self) -> torch.Tensor:
    """Cell (col, row) of the global origin, (b, 2) int64."""
    return self.get_coords(torch.zeros(3), is_global=True).squeeze(-2)

  def get_coords(self, points, is_global: bool = True) -> torch.Tensor:
    """Cells (b, n, 2) int64 of 3-D points given in global or local space."""
    pts = utils.to_tensor(points)
    if pts.dim() < 3:
      pts = pts.reshape(1, -1, 3)
    if self.proj.to_global and not is_global:
      pts = self.proj.local_to_global_space(pts)
    elif not self.proj.to_global and is_global:
      pts = self.proj.global_to_local_space(pts)
    col, row = self.proj.map_quantize(pts[..., 0], pts[..., 2])
    return torch.stack((col, row), dim=-1)

  def get_points(self, coords) -> torch.Tensor:
    """World (x, z) (b, n, 2) float32 of cells (col, row)."""
    cells = utils.to_tensor(coords)
    if cells.dim() < 3:
      cells = cells.reshape(1, -1, 2)
    x, z = self.proj.map_dequantize(cells[..., 0], cells[..., 1])
    return torch.stack((x, z), dim=-1)

  def select(self, center, crop_width: int, crop_height: int,
             fill_value: Optional[float] = None) -> "TopdownMap":
    """Crop-or-pad a window around ``center`` (b, 2) cells."""
    return crop_topdown_map(self, center=center, crop_width=crop_width,
                            crop_height=crop_height, fill_value=fill_value)

  def merge(self, *sources: "TopdownMap") -> "TopdownMap":
    """Fuse other maps into this one's frame (the reference leaves this
    unimplemented, maps.py:1951-1955)."""
    return fuse_topdown_maps(self, *sources, map_projector=self.proj)


def crop_topdown_map(source: TopdownMap, center, crop_width: int, crop_height: int,
                     fill_value: Optional[float] = None, mode: str = "nearest",
                     _validate_args: bool = True) -> TopdownMap:
  """Crop a (crop_height, crop_width) window centred at ``center`` (b, 2) and
  shift the projector's offsets accordingly (reference maps.py:1959-2037).
  Unlike the reference the caller's ``center`` tensor is not modified."""
  proj = source.proj
  center = utils.to_tensor(center).reshape(-1, 2).clone()
  dev = source.height_map.device
  woff = utils.to_tensor(proj.width_offset, device=center.device)
  hoff = utils.to_tensor(proj.height_offset, device=center.device)
  if dev.type == "cuda" and mode in F.SAMPLE_MODES:
    # one fused HIP kernel per image instead of pad + grid + grid_sample (nearest: a gather; bilinear / bicubic:
    # torch's grid_sample arithmetic, dm_crop_sample_f32)
    height_map, mask = F.crop_sample(source.height_map, center, crop_width, crop_height,
                                     fill_value=NINF, mask=source.mask, mode=mode)
    topdown = height_map
    if not source.is_height_map:
      topdown = F.crop_sample(source.topdown_map, center, crop_width, crop_height,
                              fill_value=get(fill_value, proj.fill_value), mode=mode)
  else:
    grid = utils.generate_crop_grid(center.to(dev), proj.map_width, proj.map_height,
                                    crop_width, crop_height)
    height_map = utils.image_sample(source.height_map, grid, fill_value=NINF, mode=mode)
    mask = utils.image_sample(source.mask, grid, fill_value=False, mode=mode)
    topdown = height_map
    if not source.is_height_map:
      topdown = utils.image_sample(source.topdown_map, grid,
                                   fill_value=get(fill_value, proj.fill_value), mode=mode)
  cy = center[..., 1]
  if proj.flip_h:
    cy = (proj.map_height - 1) - cy
  new_proj = proj.clone(width_offset=woff + crop_width / 2 - center[..., 0],
                        height_offset=hoff + crop_height / 2 - cy,
                        map_width=crop_width, map_height=crop_height)
  return TopdownMap(topdown_map=topdown, mask=mask, height_map=height_map,
                    is_height_map=source.is_height_map, map_projector=new_proj)


def _map_points(source: TopdownMap):
  """Cell-centre points (b, c, N, 3) of a map in GLOBAL space, its flattened
  mask and (for value maps) its flattened values (reference maps.py:2039-2069)."""
  proj = source.proj
  assert not source.is_empty and proj is not None
  cloud = proj.height_map_to_point_cloud(source.height_map)          # (b, c, h, w, 3)
  points = torch.flatten(cloud, -3, -2)
  mask = torch.flatten(source.mask.to(points.device), -2, -1)
  if proj.to_global is False:
    points = proj.local_to_global_space(points)
  values = None
  if not source.is_height_map:
    values = torch.flatten(source.topdown_map.to(points.device), -2, -1)
  return points, mask, values


def _unbroadcast_channels(t: torch.Tensor) -> torch.Tensor:
  """(b, c, h, w) tensor; a channel-broadcast view (stride 0) goes back to (b, 1, h, w)."""
  if t.dim() == 4 and t.shape[1] > 1 and t.stride(1) == 0:
    t = t[:, :1]
  return t.contiguous()


def _pose_rows(pose, b: int) -> torch.Tensor:
  p = utils.to_tensor(pose).to(torch.float32).reshape(-1, 3).cpu()
  return p.expand(b, 3) if p.shape[0] == 1 and b > 1 else p


def _offset_rows(v) -> list:
  """A map offset (python number, array or tensor, one value or one per batch row) as a list of
  float32 values held in python floats."""
  if isinstance(v, (int, float)):
    return [float(np.float32(v))]
  if torch.is_tensor(v):
    if v.device.type != "cpu" or v.dtype != torch.float32:
      v = v.detach().to(device="cpu", dtype=torch.float32)
    return v.reshape(-1).tolist()
  return np.asarray(v, dtype=np.float32).reshape(-1).tolist()


def _yaw_rows(pose, b: int, inverse: bool) -> list:
  """Per batch row the 12 floats [R | t] of local_to_global_space (rotate by yaw, then + (x, 0, z)) or, ``inverse``,
  of global_to_local_space (- (x, 0, z), then rotate by -yaw): R = utils.rotation_matrix([0, 1, 0], +-yaw) in its
  closed form (frames.py: [[d, 0, s], [0, 1, 0], [-s, 0, d]], the same float32 operations element by element;
  tests/test_host_api.py compares the two)."""
  rows = []
  for x, z, yaw in _pose_rows(pose, b).tolist():
    s, d = frames._terms(-yaw if inverse else yaw, 1)
    s, d = float(s), float(d)
    t = (-x, -0.0, -z) if inverse else (x, 0.0, z)
    rows.append((d, 0.0, s, 0.0, 1.0, 0.0, 0.0 - s, 0.0, d) + t)
  return rows


def _on(t: torch.Tensor, dev, dtype) -> torch.Tensor:
  return t if t.device == dev and t.dtype is dtype else t.to(dev, dtype)


def _fuse_source(m: TopdownMap, proj: MapProjector, dev):
  """dm_fuse_src of one source map (reference maps.py:2039-2069, 2137-2144), or None if it
  cannot go through the native path."""
  from . import _native
  sp = m.proj
  hm = _unbroadcast_channels(_on(m.height_map, dev, torch.float32))
  b, hc, h, w = hm.shape
  mk = _unbroadcast_channels(_on(m.mask, dev, torch.bool))
  val = None if m.is_height_map else _on(m.topdown_map, dev, torch.float32).contiguous()
  c = hc if val is None else val.shape[1]
  if b > _native.FUSE_MAX_BATCH or mk.shape[1] not in (1, c) or hc not in (1, c):
    return None
  if sp.map_height != h:
    return None
  src = _native.FuseSrc()
  src.height_dev, src.mask_dev = hm.data_ptr(), mk.data_ptr()
  src.value_dev = None if val is None else val.data_ptr()
  src.b, src.c, src.hc, src.mc, src.h, src.w = b, c, hc, mk.shape[1], h, w
  src.flip_h = int(bool(sp.flip_h))
  src.res, src.target_res = float(sp.map_res), float(proj.map_res)
  woff, hoff = _offset_rows(sp.width_offset), _offset_rows(sp.height_offset)
  src.woff[0:b] = woff if len(woff) == b else woff[:1] * b
  src.hoff[0:b] = hoff if len(hoff) == b else hoff[:1] * b
  src.has_l2g = int(sp.to_global is False)
  if src.has_l2g:       # local_to_global_space: rotate by yaw, then + (x, 0, z)
    for i, row in enumerate(_yaw_rows(sp.cam_pose, b, False)):
      src.l2g[i][0:12] = row
  src.has_g2l = int(proj.to_global is False)
  if src.has_g2l:       # global_to_local_space: - (x, 0, z), then rotate by -yaw
    for i, row in enumerate(_yaw_rows(proj.cam_pose, b, True)):
      src.g2l[i][0:12] = row
  return src, (hm, mk, val)      # keep the tensors alive


_STATS_SLOTS = 512
_stats_local = threading.local()     # .rings: (device, stream) -> [zero-filled (slots, 8) int32 tensor, next slot]


def _stats_host():
  """(five int32 of this host thread, their address) for dm_fuse_bbox_read_i32."""
  host = getattr(_stats_local, "host", None)
  if host is None:
    words = (ctypes.c_int32 * 5)()
    host = _stats_local.host = (words, ctypes.addressof(words))
  return host


def _zeroed_stats(dev) -> torch.Tensor:
  """Five zero words on ``dev`` for dm_fuse_bbox_multi_f32 (zero is the identity of its maxima):
  the next row of a ring that is zero-filled once per _STATS_SLOTS uses instead of once per use.
  (A row is read back -- a host sync -- before the next one is taken, so when the ring wraps no
  kernel is still writing into it.)  One ring per host thread and stream: the zero fill is
  ordered in front of its rows' uses by the stream it was enqueued on, and two threads never
  take the same row."""
  rings = getattr(_stats_local, "rings", None)
  if rings is None:
    rings = _stats_local.rings = {}
  key = (dev.type, dev.index, F._stream_ptr(dev))
  ring = rings.get(key)
  if ring is None or ring[1] >= _STATS_SLOTS:
    if len(rings) > 64:
      rings.clear()
    ring = [torch.zeros((_STATS_SLOTS, 8), dtype=torch.int32, device=dev), 0]
    rings[key] = ring
  row = ring[0][ring[1]]
  ring[1] += 1
  return row


def _fuse_topdown_maps_native(maps, proj: MapProjector, fill_value, reduction):
  """fuse_topdown_maps on dm_fuse_bbox_f32 / dm_fuse_scatter_f32 (two kernels per source
  map and one copy of five ints instead of ~80 torch kernels); None if not applicable."""
  from . import _native
  red = utils.Reduction(get(reduction, proj.reduction, "max")).value
  if red not in ("max", "min"):
    return None
  live = [m for m in maps if not m.is_empty]
  dev = live[0].height_map.device
  if dev.type != "cuda" or len({m.is_height_map for m in live}) != 1:
    return None
  is_height_map = live[0].is_height_map
  srcs = [_fuse_source(m, proj, dev) for m in live]
  if any(s is None for s in srcs):
    return None
  b, c = srcs[0][0].b, srcs[0][0].c
  hc = max(s[0].hc for s in srcs)
  if any((s[0].b, s[0].c) != (b, c) for s in srcs) or any(s[0].hc != hc for s in srcs):
    return None
  lib = _native.lib()
  stream = F._stream_ptr(dev)
  code = _native.REDUCE_MAX if red == "max" else _native.REDUCE_MIN
  # up to FUSE_MAX_SOURCES maps per launch (MapBuilder.merge: the world map and the new frame)
  groups = [srcs[i:i + _native.FUSE_MAX_SOURCES] for i in range(0, len(srcs), _native.FUSE_MAX_SOURCES)]
  arrays = [(_native.FuseSrc * len(g))(*[s[0] for s in g]) for g in groups]
  with F._on_device(dev):
    stats = _zeroed_stats(dev)
    for arr in arrays:
      _native.check(lib.dm_fuse_bbox_multi_f32(arr, len(arr), stats.data_ptr(), stream))
    # the one host sync; five maxima of order-preserving unsigned words (dm_fuse_bbox_multi_f32)
    host = _stats_host()
    _native.check(lib.dm_fuse_bbox_read_i32(stats.data_ptr(), host[1], stream))
    u = [v & 0xffffffff for v in host[0]]
    if not u[4]:
      last = maps[-1]
      return TopdownMap(topdown_map=last.topdown_map, mask=last.mask, height_map=last.height_map,
                        map_projector=proj)
    # (int32 -> ordered word: x ^ 0x80000000; the minima travel complemented)
    min_x, max_x, min_z, max_z = ((~u[0] & 0xffffffff) ^ 0x80000000, u[1] ^ 0x80000000,
                                  (~u[2] & 0xffffffff) ^ 0x80000000, u[3] ^ 0x80000000)
    min_x, max_x, min_z, max_z = (v - (1 << 32) if v >= (1 << 31) else v for v in (min_x, max_x, min_z, max_z))
    map_width = int(max_x - min_x) + 2
    map_height = int(max_z - min_z) + 2
    f32 = np.float32
    woff_v = f32(map_width / 2.) - f32(max_x + min_x) / f32(2.)
    hoff_v = f32(map_height / 2.) - f32(max_z + min_z) / f32(2.)
    fill = get(fill_value, proj.fill_value, NINF)
    topdown = torch.full((b, c, map_height, map_width), float(fill), dtype=torch.float32, device=dev)
    heights = None
    if not is_height_map:
      heights = torch.full((b, c, map_height, map_width), float(NINF), dtype=torch.float32, device=dev)
    for arr in arrays:
      _native.check(lib.dm_fuse_scatter_multi_f32(
          arr, len(arr), float(woff_v), float(hoff_v), int(bool(proj.flip_h)), map_height,
          map_width, code, topdown.data_ptr(), None if heights is None else heights.data_ptr(),
          stream))
    new_mask = F.mask_from_map(topdown, fill)
  height_map = topdown if is_height_map else heights
  woff = torch.tensor((float(woff_v),), dtype=torch.float32)
  hoff = torch.tensor((float(hoff_v),), dtype=torch.float32)
  new_proj = proj.clone(width_offset=woff, height_offset=hoff, map_width=map_width,
                        map_height=map_height)
  return TopdownMap(topdown_map=topdown, mask=new_mask, height_map=height_map,
                    map_projector=new_proj, is_height_map=is_height_map)


def fuse_topdown_maps(*maps: TopdownMap, map_projector: Optional[MapProjector] = None,
                      fill_value: Optional[float] = None,
                      reduction: Optional[Reduction] = None) -> TopdownMap:
  """Re-project several top-down maps into the frame of ``map_projector`` and
  reduce them cell-wise into one map that is just large enough (bounding box of
  the valid cells + 2), with offsets that centre it.

  Semantics of reference maps.py:2181-2287 (incl. 2071-2179): every valid cell
  becomes a point at its cell centre, goes local -> global with its own map's
  pose and global -> local with the target's, is quantised with zero offsets to
  find the bounding box (one D2H of five numbers replaces the reference's two
  ``.item()`` syncs), re-quantised with the new offsets and scattered.  All
  per-point arithmetic runs on the exact HIP primitives (dm_affine_points_f32,
  dm_map_quantize_f32, dm_scatter_f32).
  """
  if len(maps) == 0:
    return TopdownMap(map_projector=map_projector)
  if map_projector is None:
    map_projector = maps[0].proj
  proj = map_projector
  assert proj is not None, "map_projector is not provided"
  if all(m.is_empty for m in maps):
    return TopdownMap(map_projector=map_projector)
  fused = _fuse_topdown_maps_native(maps, proj, fill_value, reduction)
  if fused is not None:
    return fused
  clouds = [_map_points(m) for m in maps if not m.is_empty]
  kinds = {c[2] is None for c in clouds}
  assert len(kinds) == 1, "All maps must be the same type of maps (height or value maps)"
  is_height_map = clouds[0][2] is None
  dev = clouds[0][0].device
  points = torch.cat([c[0].to(dev) for c in clouds], dim=-2)          # (b, c, N, 3)
  masks = torch.cat([c[1].to(dev) for c in clouds], dim=-1)           # (b, c, N)
  if proj.to_global is False:
    points = proj.global_to_local_space(points)
  heights = points[..., 1]
  values = heights if is_height_map else torch.cat([c[2].to(dev) for c in clouds], dim=-1)

  # bounding box of the valid cells at zero offsets, unflipped (maps.py:2146-2179)
  col0, row0 = proj.map_quantize(points[..., 0], points[..., 2], width_offset=0.,
                                 height_offset=0., flip_h=False)
  big = torch.iinfo(torch.int64).max
  stats = torch.stack((
      torch.where(masks, col0, torch.full_like(col0, big)).amin(),
      torch.where(masks, col0, torch.full_like(col0, -big)).amax(),
      torch.where(masks, row0, torch.full_like(row0, big)).amin(),
      torch.where(masks, row0, torch.full_like(row0, -big)).amax(),
      masks.any().to(torch.int64))).cpu().tolist()                     # the one host sync
  min_x, max_x, min_z, max_z, any_valid = stats
  if not any_valid:
    last = maps[-1]
    return TopdownMap(topdown_map=last.topdown_map, mask=last.mask, height_map=last.height_map,
                      map_projector=proj)
  map_width = int(max_x - min_x) + 2
  map_height = int(max_z - min_z) + 2
  f32 = np.float32
  woff = torch.tensor([f32(map_width / 2.) - f32(max_x + min_x) / f32(2.)], dtype=torch.float32)
  hoff = torch.tensor([f32(map_height / 2.) - f32(max_z + min_z) / f32(2.)], dtype=torch.float32)

  col, row = proj.map_quantize(points[..., 0], points[..., 2], width_offset=woff,
                               height_offset=hoff, map_height=map_height)
  coords = torch.stack((row, col), dim=-1)
  canvas = torch.zeros((*values.shape[:-1], map_height, map_width), device=dev)
  topdown, new_mask = proj.project(coords=coords, values=values, masks=masks, canvas=canvas,
                                   fill_value=get(fill_value, proj.fill_value, NINF),
                                   reduction=reduction)
  if is_height_map:
    height_map = topdown
  else:
    canvas = torch.zeros((*heights.shape[:-1], map_height, map_width), device=dev)
    height_map, _ = proj.project(coords=coords, values=heights, masks=masks, canvas=canvas,
                                 fill_value=NINF, reduction=Reduction.max)
    height_map = torch.broadcast_to(height_map, topdown.shape)
  new_proj = proj.clone(width_offset=woff, height_offset=hoff, map_width=map_width,
                        map_height=map_height)
  return TopdownMap(topdown_map=topdown, mask=new_mask, height_map=height_map,
                    map_projector=new_proj, is_height_map=is_height_map)


class MapBuilder:
  """Stateful world-map builder (reference maps.py:2289-2550)."""

  def __init__(self, map_projector: MapProjector, world_map: Optional[TopdownMap] = None):
    self._proj = map_projector
    self._world_map = world_map or TopdownMap(map_projector=map_projector.clone())

  proj = property(lambda self: self._proj)
  world_map = property(lambda self: self._world_map)

  def reset(self, depth_map=None, value_map=None, valid_map=None, cam_pose=None,
            center_mode: CenterMode = CenterMode.none, **kwargs):
    """Drop the world map; optionally start a new one from a first frame."""
    self._world_map = TopdownMap(map_projector=self.proj.clone())
    if depth_map is None:
      return None
    return self.step(depth_map=depth_map, value_map=value_map, valid_map=valid_map,
                     cam_pose=cam_pose, center_mode=center_mode, **kwargs)

  def step(self, depth_map, value_map=None, valid_map=None, cam_pose=None,
           center_mode: CenterMode = CenterMode.none, merge: bool = True,
           keep_pose: bool = False, **kwargs: Dict[str, Any]) -> TopdownMap:
    """Project a frame and (optionally) merge it into the world map."""
    local = self.plot(depth_map=depth_map, value_map=value_map, valid_map=valid_map,
                      cam_pose=cam_pose, center_mode=center_mode, **kwargs)
    if merge:
      self.merge(local, keep_pose=keep_pose)
    return local

  def plot(self, depth_map, value_map=None, valid_map=None, cam_pose=None,
           center_mode: CenterMode = CenterMode.none, **kwargs: Dict[str, Any]) -> TopdownMap:
    """Project a frame to a TopdownMap; ``kwargs`` override the projector's
    orth_project defaults and are recorded in the returned map's projector."""
    cam_pose = get(cam_pose, self.proj.cam_pose, np.zeros(3, dtype=np.float32))
    woff, hoff = self._compute_offsets(cam_pose=cam_pose, center_mode=center_mode, **kwargs)
    kwargs["width_offset"], kwargs["height_offset"] = woff, hoff
    kwargs.pop("get_height_map", None)
    topdown, mask, height = self.proj.orth_project(
        depth_map=depth_map, value_map=value_map, valid_map=valid_map, cam_pose=cam_pose,
        get_height_map=True, **kwargs)
    return TopdownMap(topdown_map=topdown, mask=mask, height_map=height,
                      map_projector=self.proj.clone(cam_pose=cam_pose, **kwargs),
                      is_height_map=value_map is None)

  def merge(self, topdown_map: TopdownMap, keep_pose: bool = False,
            fill_value: Optional[float] = None,
            reduction: Optional[Reduction] = None) -> TopdownMap:
    """Fuse ``topdown_map`` into the world map."""
    if self._world_map is None:
      self._world_map = TopdownMap(map_projector=self.proj.clone())
    pose = (self._world_map if keep_pose else topdown_map).proj.cam_pose
    self._world_map = fuse_topdown_maps(
        self._world_map, topdown_map, map_projector=self.proj.clone(cam_pose=pose),
        fill_value=fill_value, reduction=reduction)
    return self._world_map

  def _compute_offsets(self, cam_pose, width_offset=None, height_offset=None, map_res=None,
                       map_width=None, map_height=None, to_global=None, center_mode=None,
                       **_unused):
    if center_mode is CenterMode.none or center_mode is None:
      # nothing to centre: the projector's own offsets or the given ones, as compute_center_offsets returns them
      # for numbers (two 0-d float32 tensors) -- without the trip through the forwarding method
      woff = self.proj.width_offset if width_offset is None else width_offset
      hoff = self.proj.height_offset if height_offset is None else height_offset
      if isinstance(woff, (int, float)) and isinstance(hoff, (int, float)) and self.proj.device is None \
          and (not torch.is_tensor(cam_pose) or cam_pose.device.type == "cpu"):
        return torch.tensor(woff, dtype=torch.float32), torch.tensor(hoff, dtype=torch.float32)
    return self.proj.compute_center_offsets(
        cam_pose=cam_pose, width_offset=width_offset, height_offset=height_offset,
        map_res=map_res, map_width=map_width, map_height=map_height, to_global=to_global,
        center_mode=center_mode)
