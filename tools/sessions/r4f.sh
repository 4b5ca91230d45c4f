#!/bin/bash
O=gpurun_out/r4f
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python tools/host_pieces.py > $O/host_pieces.log 2>&1
python tools/host_call_profile.py > $O/host_call.log 2>&1
