#!/bin/bash
# Round 4, GPU session A: tests, the re-based bench (rotating working set), kernel stats, SQ + TCC counters, stamps.
set -o pipefail
O=gpurun_out/r4a
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --rotate 1 --no-cpu-baseline --no-other-configs > $O/bench_rot1.json 2> $O/bench_rot1.err; echo "bench rot1 rc=$?"
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1; echo "sq1 rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1; echo "sq2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1; echo "write rc=$?"
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so DM_STAMPS_ROT=4 DM_STAMPS_NT=1 python tools/strip_stamps.py > $O/stamps_rot4.log 2>&1; echo "stamps rc=$?"
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so DM_STAMPS_ROT=1 DM_STAMPS_NT=1 python tools/strip_stamps.py > $O/stamps_rot1.log 2>&1; echo "stamps1 rc=$?"
ls $O
