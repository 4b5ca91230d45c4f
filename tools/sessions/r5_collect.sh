#!/bin/bash
# Round 5: profiles/r05_* from a tools/sessions/r5_final.sh run (gpurun_out/r5_final, merged back by gpurun).
# Run from the repo root with the library of that run built in tree (its md5 goes into the traffic files).
set -e
O=${1:-gpurun_out/r5_final}
P=profiles/r05
md5=$(cut -d' ' -f1 $O/lib.md5)
[ "$md5" = "$(md5sum dungeon_maps_amd/csrc/libdungeon_maps_amd.so | cut -d' ' -f1)" ] || { echo "the library in tree is not the session's ($md5)"; exit 1; }
line() { grep '^{' $1 | head -1; }
line $O/bench.json > ${P}_bench.json
line $O/bench_rot1.json > ${P}_bench_cache_resident.json
for w in cfg3 cfg4 cfg5; do line $O/bench_$w.json > ${P}_bench_$w.json; done
line $O/bench_gloo2.json > ${P}_bench_two_ranks_gloo_one_gpu.json
cp $(find $O/stats_cfg2 -name '*kernel_stats.csv' | head -1) ${P}_kernel_stats.csv
for w in cfg3 cfg4 cfg5; do cp $(find $O/stats_$w -name '*kernel_stats.csv' | head -1) ${P}_${w}_kernel_stats.csv; done
{
  echo "# Round 5 -- HBM traffic and SQ counters per kernel (PMC, MI355X), working sets beyond the Infinity Cache"
  echo
  echo "Commands: tools/sessions/r5_final.sh (one rocprofv3 pass per counter set: FETCH_SIZE, WRITE_SIZE, SQ_*; --kernel-trace --stats in a pass of its own).  Summaries by tools/pmc_summary.py."
  echo
  for w in cfg2 cfg3 cfg4 cfg5; do python3 tools/pmc_summary.py $O $w $P; done
} > ${P}_sq_counters.md
{
  echo "# cfg2 (B = 64, 640x480 -> 512x512), library md5 $md5: tools/strip_stamps.py on the -DDM_STAMPS build"
  echo "# five depth batches / output blocks in rotation (HBM-served), the streaming variant forced (the timed steps' variant)"
  grep -v amdgpu $O/stamps_rot5.log
  echo
  echo "# the same with ONE depth batch and output block (what rounds 1-3 measured)"
  grep -v amdgpu $O/stamps_rot1.log
} > ${P}_scatter_phase_stamps.log
{
  echo "# tools/skeleton.hip on the same box: the cfg2 launch's traffic with no geometry, LDS or arithmetic (speed of light of this grid)"
  echo "# (the GB/s column divides 162.5 MB by the time whatever the variant moves: read only moves 78.6 MB, write only 83.9 MB)"
  cat $O/skeleton.log
} > ${P}_traffic_skeleton.log
{ grep -v amdgpu $O/small_frame.log; grep -v amdgpu $O/host_pieces.log
  echo; echo "# tools/demo_loop_time.py: the reference demo's loop (800x600 frames, local 600x600 maps at 1.5 cm, step + merge) and cfg1's pieces"
  grep -v amdgpu $O/demo_loop.log
  echo; echo "# tools/small_frame_graph.py: DEVICE time of one cfg1 projection (HIP graph replays) by forced strips / planes"
  grep -v amdgpu $O/small_graph.log
  echo; echo "# tools/merge_profile.py: MapBuilder.merge at cfg1, piece by piece"
  grep -v amdgpu $O/merge_profile.log | head -14; } > ${P}_small_frame.log
echo "profiles/r05_* regenerated from $O (library $md5)"
{
  echo "# tools/model.hip on the same box: the traffic + issue model of the cfg2 launch (calibration, hand-off protocols, anti-phase workgroups) and the chip's rate for cfg2's read / write mix with no structure at all (mix)"
  cat $O/model.log
} > ${P}_traffic_model.log
grep -v amdgpu $O/cfg4_time.log > ${P}_cfg4_time.log
