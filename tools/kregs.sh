#!/bin/bash
# register / spill / scratch (private segment bytes) summary of the kernels in one object file: tools/kregs.sh bihome_amd/csrc/tail.o [filter]
objcopy -O binary --only-section=.hip_fatbin "$1" /tmp/kregs.fatbin && /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --input=/tmp/kregs.fatbin --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kregs.co && /opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kregs.co | python3 -c "
import sys,re
cur={}
for l in sys.stdin:
    m=re.search(r'\.(name|vgpr_count|agpr_count|vgpr_spill_count|sgpr_count|private_segment_fixed_size):\s+(\S+)',l)
    if m:
        if m.group(1) in cur:
            print('%4s v %4s a %4s spill %4s scratch  %s'%(cur.get('vgpr_count'),cur.get('agpr_count'),cur.get('vgpr_spill_count'),cur.get('private_segment_fixed_size'),cur.get('name','')[:110])); cur={}
        cur[m.group(1)]=m.group(2)
print('%4s v %4s a %4s spill %4s scratch  %s'%(cur.get('vgpr_count'),cur.get('agpr_count'),cur.get('vgpr_spill_count'),cur.get('private_segment_fixed_size'),cur.get('name','')[:110]))
" | grep "${2:-.}"
