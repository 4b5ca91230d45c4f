#!/bin/bash
# tools/kernel_regs.sh [extra hipcc flags]: registers, spills and scratch of every k_strip_scatter variant of
# dm_strip.hip as the given flags compile it (device code only: about a minute)
here=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/dm_strip_dev_$$.co
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" --cuda-device-only -c $here/dungeon_maps_amd/csrc/dm_strip.hip -o $out || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$out --output=$out.elf || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out.elf | python3 -c "
import re,sys
cur={}
for line in sys.stdin:
  m=re.match(r'\s*-?\s*\.(\w+):\s*(\S+)\s*$',line)
  if not m: continue
  k,v=m.group(1),m.group(2)
  cur[k]=v
  if k=='vgpr_spill_count':
    n=cur.get('name','')
    if 'k_strip' in n:
      print(re.sub(r'_ZN2dm12_GLOBAL__N_1\d+','',n)[:60], 'vgpr',cur.get('vgpr_count'),'spill',cur.get('vgpr_spill_count'),'sgpr_spill',cur.get('sgpr_spill_count'),'scratch',cur.get('private_segment_fixed_size'))
    cur={}
" | sort | uniq -c | sort -k1 -n | tail -${ROWS:-60}
rm -f $out $out.elf
