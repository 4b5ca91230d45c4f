#!/bin/bash
O=gpurun_out/r4i
mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_hip_full_configs.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
L=$O/campaign.log; : > $L
SEED0=3000000
run() { n=$1; shift; echo -n "$* " >> $L; env "$@" timeout -k 10 170 python tests/campaigns/parity_campaign.py $SEED0 $n 2>&1 | grep -v amdgpu | grep -E "MISMATCH|EXCEPTION|done:|Error|error" >> $L || echo "(stopped or failed)" >> $L; tail -1 $L; }
run 1500 DM_CAMPAIGN_FLOW=1
run 1200 DM_CAMPAIGN_FLOW=1 DM_CAMPAIGN_ONE_PITCH=1 DM_CAMPAIGN_SEMANTIC=0
for v in base depthnt masknt depthsc1 base; do
  if [ $v = base ]; then unset DUNGEON_MAPS_AMD_LIB; else export DUNGEON_MAPS_AMD_LIB=$PWD/tools/tmp/libdm_$v.so; fi
  python tools/launch_sweep.py fill_split=-1 2>&1 | grep fill_split | sed "s/^/$v: /"
done
