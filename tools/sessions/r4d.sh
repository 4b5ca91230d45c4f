#!/bin/bash
O=gpurun_out/r4d
mkdir -p $O
python tools/launch_sweep.py fill_split=-1 force_nt_fill=0,1 > $O/nt_fs-1.log 2>&1
python tools/launch_sweep.py fill_split=8 force_nt_fill=0,1 > $O/nt_fs8.log 2>&1
DM_ROT=1 python tools/launch_sweep.py fill_split=-1,8 > $O/rot1.log 2>&1
tail -3 $O/*.log
