"""ctypes binding of libdungeon_maps_amd.so (include/dungeon_maps_amd.h).

The HIP library IS the product: there is no eager/PyTorch or CPU fallback.  If
the shared object is missing or a symbol is absent this module raises at
import-of-use time with the build command.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DUNGEON_MAPS_AMD_LIB: load another build of the same ABI (instrumented builds)
LIB_PATH = os.environ.get("DUNGEON_MAPS_AMD_LIB") or os.path.join(
    HERE, "csrc", "libdungeon_maps_amd.so")

ABI_VERSION = 10

# dm_reduction
REDUCE_MAX, REDUCE_MIN, REDUCE_SUM, REDUCE_MEAN, REDUCE_PROD = range(5)

FRAME_FLOATS = 32  # sizeof(dm_frame) / 4


class Params(ctypes.Structure):
  """dm_params"""
  _fields_ = [(n, ctypes.c_int32) for n in (
      "B", "dc", "vc", "H", "W", "mh", "mw", "clip_border", "flip_h", "to_global",
      "reduction", "has_dmin", "has_dmax", "has_hmax", "valid_c")] + [
      (n, ctypes.c_float) for n in (
          "cx", "cy", "fx", "fy", "res", "fill", "dmin", "dmax", "hmax")]


FUSE_MAX_BATCH = 8
FUSE_MAX_SOURCES = 4


class FuseSrc(ctypes.Structure):
  """dm_fuse_src"""
  _fields_ = [("height_dev", ctypes.c_void_p), ("mask_dev", ctypes.c_void_p),
              ("value_dev", ctypes.c_void_p)] + [
      (n, ctypes.c_int32) for n in ("b", "c", "hc", "mc", "h", "w", "flip_h", "has_l2g",
                                    "has_g2l")] + [
      ("res", ctypes.c_float), ("target_res", ctypes.c_float),
      ("woff", ctypes.c_float * FUSE_MAX_BATCH), ("hoff", ctypes.c_float * FUSE_MAX_BATCH),
      ("l2g", (ctypes.c_float * 12) * FUSE_MAX_BATCH),
      ("g2l", (ctypes.c_float * 12) * FUSE_MAX_BATCH)]


class FramesPlan(ctypes.Structure):
  """dm_frames_plan"""
  _fields_ = [(n, ctypes.c_int32) for n in (
      "strips", "strip_width", "slab_cells", "max_rows", "max_union_cells", "slack_cells",
      "magnitude")] + [("pitch", ctypes.c_float * 4), ("reserved", ctypes.c_int32)]


# dm_status_bits
STATUS_FRAME_DID_NOT_FIT, STATUS_LIST_OVERFLOW = 0x1, 0x100


class NativeError(RuntimeError):
  pass


_SIGNATURES = {
    "dm_version": (ctypes.c_int, []),
    "dm_build_flags": (ctypes.c_char_p, []),
    "dm_last_error": (ctypes.c_char_p, []),
    "dm_orth_project_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Params)]),
    "dm_orth_project_f32": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
        ctypes.c_void_p]),
    "dm_frames_fill_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p]),
    "dm_frames_prepared_bytes": (ctypes.c_size_t, [ctypes.POINTER(Params)]),
    "dm_frames_prepare_f32": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
        ctypes.POINTER(FramesPlan), ctypes.POINTER(FramesPlan), ctypes.c_void_p]),
    "dm_orth_project_prepared_f32": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.POINTER(FramesPlan), ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
        ctypes.c_void_p]),
    "dm_orth_project_fused_f32": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_orth_project_flow_f32": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_debug_last_flow_fused": (ctypes.c_int, []),
    "dm_debug_flow_fused": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_force_generic_path": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_force_bands": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_slab_budget": (ctypes.c_size_t, [ctypes.c_size_t]),
    "dm_debug_count_escapes": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_debug_last_path": (ctypes.c_int, []),
    "dm_debug_force_legacy_window": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_force_strips": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_planes": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_fill_split": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_force_nt_fill": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_strip_value_list": (ctypes.c_int, [ctypes.c_int]),
    "dm_debug_strip_slab_budget": (ctypes.c_size_t, [ctypes.c_size_t]),
    "dm_debug_last_strip_info": (None, [ctypes.POINTER(ctypes.c_int32)]),
    "dm_debug_force_fused_split": (None, [ctypes.c_int, ctypes.c_int]),
    "dm_debug_last_fused_split": (None, [ctypes.POINTER(ctypes.c_int32)]),
    "dm_debug_strip_geometry": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_debug_strip_geometry_dev": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
        ctypes.c_void_p]),
    "dm_debug_last_split": (None, [ctypes.POINTER(ctypes.c_int32)]),
    "dm_debug_windows": (ctypes.c_int, [ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                        ctypes.c_size_t]),
    "dm_fuse_bbox_f32": (ctypes.c_int, [
        ctypes.POINTER(FuseSrc), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "dm_fuse_scatter_f32": (ctypes.c_int, [
        ctypes.POINTER(FuseSrc), ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_fuse_bbox_multi_f32": (ctypes.c_int, [
        ctypes.POINTER(FuseSrc), ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_fuse_bbox_read_i32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_fuse_scatter_multi_f32": (ctypes.c_int, [
        ctypes.POINTER(FuseSrc), ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_int,
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_crop_nearest_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_crop_sample_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_debug_record_after_projection": (None, [ctypes.c_void_p]),
    "dm_debug_record_before_projection": (None, [ctypes.c_void_p]),
    "dm_camera_affine_grid_f32": (ctypes.c_int, [
        ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dm_affine_points_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t,
        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "dm_map_quantize_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_size_t, ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p]),
    "dm_scatter_workspace_bytes": (ctypes.c_size_t, [
        ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32]),
    "dm_scatter_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_float,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dm_fuse_batch_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]),
    "dm_mask_from_map_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t,
        ctypes.c_void_p]),
}

_lib = None


def exported_symbols():
  """Names include/dungeon_maps_amd.h and include/dungeon_maps_amd_debug.h declare (checked by the CPU tests)."""
  return sorted(_SIGNATURES)


def lib():
  """Load the shared library once; fail loudly when it is not there."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise NativeError(
        f"{LIB_PATH} is missing: the HIP extension is the only compute path of "
        "dungeon_maps_amd (no CPU/eager fallback).  Build it with "
        "`make -C dungeon_maps_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`.")
  try:
    handle = ctypes.CDLL(LIB_PATH)
  except OSError as e:
    raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
  for name, (restype, argtypes) in _SIGNATURES.items():
    try:
      fn = getattr(handle, name)
    except AttributeError as e:
      raise NativeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
    fn.restype = restype
    fn.argtypes = argtypes
  if handle.dm_version() != ABI_VERSION:
    raise NativeError(
        f"ABI mismatch: library {handle.dm_version()} vs binding {ABI_VERSION}; rebuild")
  _lib = handle
  return _lib


def check(rc):
  if rc != 0:
    msg = lib().dm_last_error()
    raise NativeError(f"dungeon_maps_amd native call failed ({rc}): "
                      f"{msg.decode() if msg else 'unknown error'}")


# ---------------------------------------------------------------------------
# Status word (dm_status_bits): one int32 of pinned host memory per process.  The kernels store
# into it when they refuse a frame the host could not refuse up front (a prepared batch whose
# device-side pose records changed behind its plan's back); the host looks at it -- a plain
# memory read, no synchronisation -- in front of every projection call and raises.
# ---------------------------------------------------------------------------
_status = None


def status_word():
  """(pinned int32 tensor of one element, its address); allocated on first use."""
  global _status
  if _status is None:
    import torch
    t = torch.zeros(1, dtype=torch.int32).pin_memory()
    _status = (t, t.numpy(), t.data_ptr())
  return _status


def status_ptr() -> int:
  return status_word()[2]


def check_status():
  """Raise if an earlier projection flagged a frame (the flag is cleared by the raise)."""
  if _status is None:
    return
  bits = int(_status[1][0])
  if bits:
    _status[1][0] = 0
    what = []
    if bits & STATUS_FRAME_DID_NOT_FIT:
      what.append("a frame's camera state did not fit the launch plan it was projected with "
                  "(a prepared batch's device buffer changed behind its plan): its maps hold the fill value")
    if bits & STATUS_LIST_OVERFLOW:
      what.append("a strip's shared-group list overflowed")
    raise NativeError("an earlier projection reported: " + "; ".join(what or [f"status {bits}"]))
