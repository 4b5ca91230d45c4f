#!/bin/bash
O=gpurun_out/r4g
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_hip_rows_next.py tests/test_hip_full_configs.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg4 --rotate 1 --no-cpu-baseline > $O/bench_cfg4_rot1.json 2> $O/bench_cfg4_rot1.err; echo "cfg4 rot1 rc=$?"
B4="python3 bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg4 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- $B4 > $O/stats_cfg4.log 2>&1; echo "stats cfg4 rc=$?"
B3="python3 bench.py --gpus 1 --steps 4 --warmup 2 --workload cfg3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -- $B3 > $O/stats_cfg3.log 2>&1; echo "stats cfg3 rc=$?"
B5="python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload cfg5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -- $B5 > $O/stats_cfg5.log 2>&1; echo "stats cfg5 rc=$?"
for w in cfg3 cfg4 cfg5; do
  case $w in cfg3) BB=$B3;; cfg4) BB=$B4;; cfg5) BB=$B5;; esac
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$w -- $BB > $O/fetch_$w.log 2>&1; echo "fetch $w rc=$?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$w -- $BB > $O/write_$w.log 2>&1; echo "write $w rc=$?"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_$w -- $BB > $O/sq_$w.log 2>&1; echo "sq $w rc=$?"
done
