#!/bin/bash
O=gpurun_out/r4j
mkdir -p $O
for v in base depthnt valuent bothnt; do
  if [ $v = base ]; then unset DUNGEON_MAPS_AMD_LIB; else export DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_$v.so; fi
  for w in cfg2 cfg3 cfg4 cfg5; do
    steps=20; [ $w = cfg3 ] && steps=6
    python bench.py --gpus 1 --steps $steps --warmup 4 --workload $w --no-cpu-baseline --no-other-configs > $O/${v}_$w.json 2> $O/${v}_$w.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/${v}_$w.json") if l.startswith("{")][0])
extra = ""
if "legs" in d: extra = " legs: " + ", ".join(f"{k} {x['us']:.1f}" for k, x in d["legs"].items())
print("$v $w: launch %.2f us  frac %.3f  ms/step %.4f%s" % (d["roofline"]["launch_us"], d["roofline"]["frac"], d["ms_per_step"], extra))
PY
  done
done
