"""Static check of the built library's device code (no GPU needed) for a store hazard of gfx950
that the compiler's hazard recogniser does not cover.

A buffer store of more than 8 bytes keeps reading its data registers for a few cycles after it
has issued; LLVM inserts the wait states in front of a vector instruction that overwrites them,
EXCEPT when the store's scalar-offset field holds a register (it takes that addressing form to be
free of the hazard).  On MI355X it is not: round 3's parity campaign found the fill duty's
buffer_store_dwordx4 -- scalar offset in an SGPR -- storing an image-row number in place of the
fill value after the very next instruction reused its first data register (DESIGN.md, section 7;
the fix: dm_pixel.hpp buffer_store_b128_at_scalar_offset).  This test disassembles every gfx950
code object of the library and requires two wait states between any such store and a vector
instruction that writes one of its data registers."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from dungeon_maps_amd import _native

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WIDE_STORE = re.compile(r"^buffer_store_(dwordx3|dwordx4|format_xyz|format_xyzw)\b")
REGS = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")
REQUIRED_WAIT_STATES = 2


def _regs(operand):
  m = REGS.match(operand.strip())
  if not m:
    return set()
  if m.group(1) is not None:
    return {int(m.group(1))}
  return set(range(int(m.group(2)), int(m.group(3)) + 1))


def _instructions(path):
  """(kernel symbol, mnemonic, [operands]) of every instruction of a code object, in order."""
  text = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True, check=True).stdout
  kernel = None
  out = []
  for line in text.splitlines():
    if line.endswith(">:"):
      kernel = line.split("<", 1)[1][:-2]
      continue
    if not line.startswith("\t"):
      continue
    body = line.split("//", 1)[0].strip()
    if not body:
      continue
    parts = body.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    out.append((kernel, parts[0], ops))
  return out


def hazards(instructions):
  """Wide buffer stores with a register in the scalar-offset field whose data registers a vector
  instruction overwrites within REQUIRED_WAIT_STATES wait states."""
  found = []
  for i, (kernel, mnem, ops) in enumerate(instructions):
    if not WIDE_STORE.match(mnem) or len(ops) < 4:
      continue
    soffset = ops[3].split()[0]            # "s4 offen" -> "s4"
    if not re.match(r"^s\d+$", soffset):   # 0 / off / a literal: the compiler handles those itself
      continue
    data = _regs(ops[0])
    waited = 0
    for kernel2, m2, ops2 in instructions[i + 1:i + 1 + 8]:
      if waited >= REQUIRED_WAIT_STATES or kernel2 != kernel:
        break
      if m2.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
        break                              # (control flow: not followed)
      if m2.startswith("v_") and ops2 and not m2.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        if _regs(ops2[0]) & data:
          found.append((kernel, mnem + " " + ", ".join(ops), m2 + " " + ", ".join(ops2), waited))
      waited += int(ops2[0], 0) + 1 if m2 == "s_nop" and ops2 else 1
  return found


def test_the_checker_sees_the_hazard_and_its_fix():
  bad = [("k", "buffer_store_dwordx4", ["v[22:25]", "v1", "s[36:39]", "s12 offen"]),
         ("k", "v_cvt_f32_i32_e32", ["v22", "v17"])]
  assert len(hazards(bad)) == 1
  fixed = [bad[0], ("k", "v_cndmask_b32_e32", ["v1", "v31", "v96", "vcc"]), ("k", "s_nop", ["1"]), bad[1]]
  assert hazards(fixed) == []
  immediate = [("k", "buffer_store_dwordx4", ["v[22:25]", "v1", "s[36:39]", "0 offen"]), bad[1]]
  assert hazards(immediate) == []          # (the compiler's own wait states cover this form)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="needs the ROCm llvm-objdump")
def test_no_wide_store_with_a_scalar_offset_register_has_its_data_overwritten_in_flight():
  _native.lib()                            # the library must be there: no fallback
  tmp = tempfile.mkdtemp(prefix="dm_hazard_")
  try:
    lib = shutil.copy(_native.LIB_PATH, tmp)
    subprocess.run([OBJDUMP, "--offloading", lib], capture_output=True, text=True, check=True, cwd=tmp)
    objects = sorted(f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f)
    assert objects, "no gfx950 code object in the library"
    stores = 0
    found = []
    for f in objects:
      ins = _instructions(os.path.join(tmp, f))
      stores += sum(1 for _, m, ops in ins if WIDE_STORE.match(m) and len(ops) > 3
                    and re.match(r"^s\d+$", ops[3].split()[0]))
      found += hazards(ins)
    assert stores > 0, "the fill duties' stores were not found: has the disassembly's format changed?"
    assert not found, "\n".join("%s: %s  <-  %s  (after %d wait states)" % h for h in found[:10])
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
