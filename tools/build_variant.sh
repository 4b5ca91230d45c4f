#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags]: an experimental build of the library
# (dm_window.hip, dm_strip.hip and dm_points.hip recompiled with the flags) -> tools/tmp/libdm_NAME.so; select it with
# DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_NAME.so
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/dungeon_maps_amd/csrc
mkdir -p $here/tools/tmp
make -C $src >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  "$@" -c $src/dm_window.hip -o /tmp/dm_window_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  "$@" -c $src/dm_strip.hip -o /tmp/dm_strip_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  "$@" -c $src/dm_points.hip -o /tmp/dm_points_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $here/tools/tmp/libdm_$name.so \
  $src/dm_api.o $src/dm_generic.o /tmp/dm_window_$name.o /tmp/dm_strip_$name.o /tmp/dm_points_$name.o
echo $here/tools/tmp/libdm_$name.so
