#!/bin/bash
# tools/ab_libs.sh SWITCH=VALS LIB [LIB ...]: the cfg2 launch sequence / step (tools/launch_sweep.py) with each of the
# given library builds in turn, twice over, in ONE process sequence on ONE box (boxes of the pool differ by +-3 %).
# LIB: a path, or "tree" for the in-tree build.
sw=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = tree ]; then unset DUNGEON_MAPS_AMD_LIB; else export DUNGEON_MAPS_AMD_LIB=$lib; fi
    echo "== $lib"
    python $here/tools/launch_sweep.py $sw 2>&1 | grep -v amdgpu.ids
  done
done
