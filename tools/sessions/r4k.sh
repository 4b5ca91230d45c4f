#!/bin/bash
for v in base flushnt masknt flushmasknt base; do
  if [ $v = base ]; then unset DUNGEON_MAPS_AMD_LIB; else export DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_$v.so; fi
  python tools/launch_sweep.py fill_split=-1 2>&1 | grep fill_split | sed "s/^/$v: /"
done
DM_ROT=1 python tools/launch_sweep.py fill_split=-1 2>&1 | grep fill_split | sed "s/^/base rot1: /"
