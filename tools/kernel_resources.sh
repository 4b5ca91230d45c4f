#!/bin/bash
# tools/kernel_resources.sh [file.hip ...]: registers, spills, scratch and occupancy of every kernel
# (device-only compile with -Rpass-analysis=kernel-resource-usage; default: the strip path).
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/dungeon_maps_amd/csrc
files=${@:-dm_strip.hip}
for f in $files; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
    --offload-device-only -Rpass-analysis=kernel-resource-usage -c $src/$f -o /tmp/kres_$$.co 2>&1 |
  grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: //' |
  paste - - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[ \t]\+/ /g; s/_ZN2dm12_GLOBAL__N_1[0-9]*//'
done
rm -f /tmp/kres_$$.co
