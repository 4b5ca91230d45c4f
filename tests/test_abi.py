"""CPU checks of the drop-in boundary: the C-ABI shared library loads and
exports every symbol include/dungeon_maps_amd.h declares (no compute calls --
there is no GPU in the build container), struct layouts match the header, and
the product refuses to run without its HIP library or without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dungeon_maps_amd.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "dungeon_maps_amd_debug.h")


def _declared_functions(*headers):
  found = set()
  for h in headers or (HEADER, DEBUG_HEADER):
    src = open(h).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    found |= set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", src))
  return sorted(found)


def test_debug_hooks_live_in_their_own_header():
  """The drop-in boundary declares no test / measurement hook; the debug header declares nothing else."""
  assert not [n for n in _declared_functions(HEADER) if n.startswith("dm_debug_")]
  assert all(n.startswith("dm_debug_") for n in _declared_functions(DEBUG_HEADER))


def test_library_exports_every_declared_symbol():
  from dungeon_maps_amd import _native
  lib = _native.lib()
  declared = _declared_functions()
  assert declared, "no functions parsed from the header"
  assert set(declared) == set(_native.exported_symbols())
  for name in declared:
    assert hasattr(lib, name), name
  assert lib.dm_version() == _native.ABI_VERSION


def test_struct_layouts_match_header():
  from dungeon_maps_amd import _native
  assert ctypes.sizeof(_native.Params) == 15 * 4 + 9 * 4
  assert _native.FRAME_FLOATS * 4 == 128          # sizeof(dm_frame)
  src = open(HEADER).read()
  body = re.search(r"typedef struct dm_params \{(.*?)\} dm_params;", src, re.S).group(1)
  body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
  names = []
  for decl in body.split(";"):
    decl = decl.strip()
    if decl:
      names += [n.strip() for n in decl.split(None, 1)[1].split(",")]
  assert names == [f[0] for f in _native.Params._fields_]


def test_argument_validation_without_gpu():
  """Argument errors are reported before any HIP call is made."""
  from dungeon_maps_amd import _native
  lib = _native.lib()
  p = _native.Params()
  p.B, p.dc, p.vc, p.H, p.W, p.mh, p.mw = 1, 2, 3, 4, 4, 8, 8   # dc not in (1, vc)
  rc = lib.dm_orth_project_f32(ctypes.byref(p), 1, 1, 1, None, 1, 1, None, None, None, None, 0, None, None)
  assert rc == -1 and b"depth channels" in lib.dm_last_error()
  p.dc, p.vc, p.reduction = 1, 0, 9
  rc = lib.dm_orth_project_f32(ctypes.byref(p), 1, 1, None, None, 1, 1, None, None, None, None, 0, None, None)
  assert rc == -1 and b"reduction" in lib.dm_last_error()
  p.reduction = 2
  rc = lib.dm_orth_project_fused_f32(ctypes.byref(p), 1, 1, None, None, 1, 1, 0, None, 0, None, None)
  assert rc == -2 and b"max/min" in lib.dm_last_error()
  with pytest.raises(_native.NativeError):
    _native.check(rc)


def test_no_cpu_fallback():
  import torch
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  import dungeon_maps_amd as dmap
  proj = dmap.MapProjector(width=8, height=4, hfov=1.2, cam_pitch=0., cam_height=1.,
                           width_offset=4., height_offset=4., map_res=0.5, map_width=8,
                           map_height=8)
  with pytest.raises(RuntimeError, match="no CPU"):
    proj.orth_project(np.ones((4, 8), dtype=np.float32))


def test_product_does_not_import_the_oracle():
  pkg = os.path.join(ROOT, "dungeon_maps_amd")
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
        text = open(os.path.join(dirpath, f)).read()
        assert "oracle" not in text.lower(), f"{f} mentions the oracle"
  # the measurement / profiling helpers under tools/ do not use it either (the parity campaigns
  # that do live under tests/campaigns/)
  tools = os.path.join(ROOT, "tools")
  for f in sorted(os.listdir(tools)):
    if f.endswith((".py", ".sh", ".hip", ".c")):
      text = open(os.path.join(tools, f)).read()
      assert "import oracle" not in text and "from oracle" not in text, f"tools/{f} uses the oracle"
