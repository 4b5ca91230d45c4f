#!/bin/bash
# Round 5 parity campaigns on the final library (builder-run; profiles/r05_parity_campaigns.log)
O=gpurun_out/r5_campaign
rm -rf $O
mkdir -p $O
L=$O/campaign.log
echo "# library md5 $(md5sum dungeon_maps_amd/csrc/libdungeon_maps_amd.so | cut -d' ' -f1); seeds from ${SEED0:=5000000}" > $L
run() { n=$1; shift; echo -n "$* " >> $L; env "$@" timeout -k 10 ${LIMIT:-170} python tests/campaigns/parity_campaign.py $SEED0 $n 2>&1 | grep -v amdgpu | grep -E "MISMATCH|EXCEPTION|done:|Error|error" >> $L || echo "(stopped by its time limit or failed)" >> $L; tail -1 $L; }
run 1500 DM_X=0
run 1500 DM_CAMPAIGN_ONE_PITCH=1 DM_CAMPAIGN_CALLS=1
run 1500 DM_CAMPAIGN_FILL_SPLIT=1 DM_CAMPAIGN_ONE_PITCH=1 DM_CAMPAIGN_CALLS=1
run 1500 DM_CAMPAIGN_FLOW=1
run 1200 DM_CAMPAIGN_FLOW=1 DM_CAMPAIGN_ONE_PITCH=1 DM_CAMPAIGN_SEMANTIC=0
run 500 DM_CAMPAIGN_FLOW=1 DM_CAMPAIGN_BIG=1 DM_CAMPAIGN_SEMANTIC=0
run 600 DM_CAMPAIGN_FILL_SPLIT=1 DM_CAMPAIGN_BIG=1 DM_CAMPAIGN_CALLS=1
run 1000 DM_CAMPAIGN_FUSED=1
run 800 DM_CAMPAIGN_SUM=1
run 800 DM_CAMPAIGN_SUM=mean
run 800 DM_CAMPAIGN_ODD=1
run 800 DM_CAMPAIGN_OFFSETS=1 DM_CAMPAIGN_ONE_PITCH=1 DM_CAMPAIGN_CALLS=1
run 800 DM_CAMPAIGN_DC=1 DM_CAMPAIGN_CALLS=1
echo -n "crop campaign " >> $L; timeout -k 10 170 python tests/campaigns/crop_campaign.py $SEED0 1500 2>&1 | grep -E "MISMATCH|done:|Error|error" >> $L || echo "(stopped by its time limit or failed)" >> $L; tail -1 $L
echo -n "fuse campaign " >> $L; timeout -k 10 170 python tests/campaigns/fuse_campaign.py $SEED0 ${FUSE_N:-600} 2>&1 | grep -E "MISMATCH|EXCEPTION|done:|Error|error" >> $L || echo "(stopped by its time limit or failed)" >> $L; tail -1 $L
