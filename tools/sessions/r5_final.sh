#!/bin/bash
# Round 5: the measurements the committed profiles/r04_* files come from (one gpurun call).
O=gpurun_out/r5_final
rm -rf $O
mkdir -p $O
export TMPDIR=/tmp
md5sum dungeon_maps_amd/csrc/libdungeon_maps_amd.so > $O/lib.md5
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --rotate 1 --no-cpu-baseline --no-other-configs > $O/bench_rot1.json 2> $O/bench_rot1.err; echo "bench rot1 rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python bench.py --gpus 1 --steps 6 --warmup 2 --workload cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
DM_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29537 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
B2="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-two-streams"
B3="python3 bench.py --gpus 1 --steps 4 --warmup 2 --workload cfg3 --no-cpu-baseline --no-two-streams"
B4="python3 bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg4 --no-cpu-baseline"
B5="python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload cfg5 --no-cpu-baseline"
for w in cfg2 cfg3 cfg4 cfg5; do
  case $w in cfg2) BB=$B2;; cfg3) BB=$B3;; cfg4) BB=$B4;; cfg5) BB=$B5;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- $BB > $O/stats_$w.log 2>&1; echo "stats $w rc=$?"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$w -- $BB > $O/fetch_$w.log 2>&1; echo "fetch $w rc=$?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$w -- $BB > $O/write_$w.log 2>&1; echo "write $w rc=$?"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_$w -- $BB > $O/sq_$w.log 2>&1; echo "sq $w rc=$?"
done
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so DM_STAMPS_ROT=5 DM_STAMPS_NT=1 python tools/strip_stamps.py > $O/stamps_rot5.log 2>&1; echo "stamps rc=$?"
DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_stamps.so DM_STAMPS_ROT=1 DM_STAMPS_NT=1 python tools/strip_stamps.py > $O/stamps_rot1.log 2>&1; echo "stamps1 rc=$?"
tools/tmp/skeleton > $O/skeleton.log 2>&1; echo "skeleton rc=$?"
tools/tmp/model all > $O/model.log 2>&1; echo "model rc=$?"
python tools/cfg4_time.py > $O/cfg4_time.log 2>&1; echo "cfg4 time rc=$?"
python tools/small_frame_time.py > $O/small_frame.log 2>&1; echo "small rc=$?"
python tools/host_pieces.py > $O/host_pieces.log 2>&1; echo "host rc=$?"
python tools/demo_loop_time.py > $O/demo_loop.log 2>&1; echo "demo rc=$?"
python tools/small_frame_graph.py > $O/small_graph.log 2>&1; echo "graph rc=$?"
python tools/merge_profile.py > $O/merge_profile.log 2>&1; echo "merge rc=$?"
