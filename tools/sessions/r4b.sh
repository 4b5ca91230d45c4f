#!/bin/bash
set -o pipefail
O=gpurun_out/r4b
mkdir -p $O
python -m pytest tests/test_hip_strip.py tests/test_hip_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
python tools/launch_sweep.py fill_split=-1,0,2,4,6,8 > $O/sweep.log 2>&1; echo "sweep rc=$?"
tail -8 $O/sweep.log
