#!/bin/bash
set -o pipefail
O=gpurun_out/r4c
mkdir -p $O
for v in stamps stamps_nomath stamps_nofill; do
  for fs in -1 8; do
    DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_$v.so DM_STAMPS_ROT=4 DM_STAMPS_NT=1 DM_STAMPS_FILL_SPLIT=$fs python tools/strip_stamps.py > $O/${v}_fs$fs.log 2>&1; echo "$v fs=$fs rc=$?"
  done
done
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so DM_STAMPS_ROT=1 DM_STAMPS_NT=1 DM_STAMPS_FILL_SPLIT=-1 python tools/strip_stamps.py > $O/stamps_rot1.log 2>&1
