#!/bin/bash
for i in 1 2 3; do
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so DM_STAMPS_ROT=5 DM_STAMPS_NT=1 DM_STAMPS_SLOWEST=14 python tools/strip_stamps.py 2>&1 | grep -A16 "kernel span\|slowest workgroups" | grep -v "loop end\|first / last\|pixel loop by\|slowest frames\|fastest frames\|per-WG\|corr"
done
