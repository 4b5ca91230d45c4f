#!/bin/bash
O=gpurun_out/r4m
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/launch_sweep.py fill_split=-1 2>&1 | grep fill_split
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so DM_STAMPS_ROT=5 DM_STAMPS_NT=1 DM_STAMPS_SLOWEST=6 python tools/strip_stamps.py 2>&1 | grep -v amdgpu | grep -A8 "kernel span\|slowest workgroups\|row tables\|loop scalars" | head -40
python tools/launch_sweep.py fill_split=-1 2>&1 | grep fill_split
