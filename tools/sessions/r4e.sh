#!/bin/bash
O=gpurun_out/r4e
mkdir -p $O
python tools/launch_sweep.py fill_split=-1 > $O/base.log 2>&1
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_noprio.so python tools/launch_sweep.py fill_split=-1 > $O/noprio.log 2>&1
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_rotprio.so python tools/launch_sweep.py fill_split=-1 > $O/rotprio.log 2>&1
python tools/launch_sweep.py fill_split=-1 > $O/base2.log 2>&1
grep -h fill_split $O/*.log
