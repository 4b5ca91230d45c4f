"""Static check of the built library's kernels (no GPU needed): the hot kernels keep everything in registers.

The projection kernels are capped at 128 VGPRs by their 1024 threads; a change that tips one of them over the
edge compiles, passes every parity test and runs with its spilled registers in scratch memory -- this round's
fused-flow variant did exactly that (80 VGPR spills) until its lambda was forced inline.  The code objects'
own metadata says so: every kernel on a measured path must report no scratch and no VGPR spills.  (SGPR
spills into vector lanes are tolerated: the compiler keeps them out of the pixel loops -- tools)
Writing this test found the window kernel's value-map variants spilling 21-95 registers (fixed: two image rows
in flight).  One kernel is left and listed: the general-rotation / IEEE-division fallback (FAST = false) of the
mean reduction with a valid map spills ONE register; no projector call reaches it.  Every other kernel of the
library, hot or not, must be clean."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from dungeon_maps_amd import _native

LLVM = "/opt/rocm/lib/llvm/bin"
HOT = ("k_strip_scatter", "k_strip_combine", "k_strip_combine_planes", "k_strip_fused", "k_fuse_unions", "k_fuse_windows", "k_window_merge",
       "k_camera_affine_grid4", "k_crop_nearest4")


KNOWN_TO_SPILL = r"k_window_scatterILi3ELb0ELb1ELb0E"      # <mean, FAST = false, HAS_VALID, no values>: 1 VGPR


def _kernels(path):
  """{mangled name: {private_segment_fixed_size, vgpr_spill_count, vgpr_count}} of a code object."""
  notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True,
                         check=True).stdout
  out, cur = {}, None
  fields = {}
  for line in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)\s*$", line)
    if not m:
      continue
    key, val = m.group(1), m.group(2)
    if key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "name", "symbol"):
      fields[key] = val
    if key == "vgpr_spill_count":          # (the last of a kernel's fields we read)
      if "name" in fields:
        out[fields["name"]] = {k: int(v) for k, v in fields.items() if k not in ("name", "symbol")}
      fields = {}
  return out


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="needs the ROCm llvm tools")
def test_kernels_use_no_scratch_and_spill_no_vector_registers():
  _native.lib()                            # the library must be there: no fallback
  tmp = tempfile.mkdtemp(prefix="dm_codegen_")
  try:
    lib = shutil.copy(_native.LIB_PATH, tmp)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], capture_output=True, text=True,
                   check=True, cwd=tmp)
    objects = sorted(f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f)
    assert objects, "no gfx950 code object in the library"
    seen, bad = set(), []
    for f in objects:
      for name, k in _kernels(os.path.join(tmp, f)).items():
        for h in HOT:                      # (the kernels of the measured paths must be there at all)
          if h in name:
            seen.add(h)
        if re.search(r"k_window_scatterILi\dELb1E", name):      # the window kernel's FAST variants (template parameter 2)
          seen.add("k_window_scatter<., FAST>")
        if re.search(KNOWN_TO_SPILL, name):
          continue
        if k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0):
          bad.append((name, k))
        assert k.get("vgpr_count", 0) <= 128 or "k_strip" not in name, (name, k)
    want = set(HOT) | {"k_window_scatter<., FAST>"}
    assert seen == want, "kernels not found in the library's metadata: %s" % sorted(want - seen)
    assert not bad, "\n".join("%s: %s" % b for b in bad)
  finally:
    shutil.rmtree(tmp, ignore_errors=True)


def test_shipped_library_is_the_product_build():
  """No measurement / instrumentation flag in the library the package loads (dm_build_flags): timing-only
  builds that compute wrong maps do not exist in this tree (tools/experiments keeps them as patches, the
  sources #error on their macros), instrumented ones (-DDM_STAMPS) live under tools/tmp."""
  lib = _native.lib()
  assert lib.dm_build_flags() == b"", lib.dm_build_flags()
  src = os.path.join(os.path.dirname(_native.LIB_PATH))
  hits = []
  for f in sorted(os.listdir(src)):
    if f.endswith((".hip", ".hpp")):
      for i, line in enumerate(open(os.path.join(src, f)), 1):
        if re.search(r"#\s*if.*DM_X_", line) and "#error" not in line and "defined(DM_X_NO" not in line:
          hits.append(f"{f}:{i}: {line.strip()}")
  assert not hits, "measurement switches inside the product sources:\n" + "\n".join(hits)
