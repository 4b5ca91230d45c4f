#!/bin/bash
O=gpurun_out/r4h
mkdir -p $O
for v in base fph1 fph3 cs16 cs64 base; do
  if [ $v = base ]; then unset DUNGEON_MAPS_AMD_LIB; else export DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_$v.so; fi
  python tools/launch_sweep.py fill_split=-1 2>&1 | grep fill_split | sed "s/^/$v: /"
done
