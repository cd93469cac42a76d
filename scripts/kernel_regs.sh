#!/bin/bash
# register / scratch / LDS budget of every kernel in an object of xview2_amd/build (no GPU needed):
#   scripts/kernel_regs.sh igemm_conv [filter]
# prints  name  vgpr  agpr  scratch_bytes  lds_bytes  per kernel, from the code object's metadata notes
set -e
tmp=$(mktemp -d)
cp xview2_amd/build/$1.o $tmp/k.o
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading k.o >/dev/null 2>&1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | python3 -c "
import sys,re
txt=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for blk in re.split(r'\n\s*- \.agpr_count:', txt)[1:]:
    blk='.agpr_count:'+blk
    g=lambda k:(re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if flt in name: print(name[:120], 'vgpr',g('vgpr_count'),'agpr',g('agpr_count'),'scratch',g('private_segment_fixed_size'),'lds',g('group_segment_fixed_size'))
" "$2"
rm -rf $tmp
