#!/bin/bash
# disassemble ONE kernel of an object under xview2_amd/build:  scripts/kisa.sh OBJECT(.o name without dir) MANGLED_PREFIX OUT.s
# prints the register budget of that kernel
set -e
tmp=$(mktemp -d)
cp $(dirname $0)/../xview2_amd/build/$1 $tmp/k.o
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading k.o >/dev/null 2>&1)
CO=$tmp/k.o.0.hipv4-amdgcn-amd-amdhsa--gfx950
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO | grep -A12 "\.name: *$2" | grep "vgpr_count\|private_segment_fixed\|\.sgpr_count" | head -3
/opt/rocm/lib/llvm/bin/llvm-objdump -d $CO > $tmp/all.s
L=$(grep -n "^[0-9a-f]* <$2" $tmp/all.s | head -1 | cut -d: -f1)
awk -v L=$L 'NR>=L' $tmp/all.s | awk '/^[0-9a-f]+ <_ZN/ && NR>1 {exit} {print}' > $3
wc -l $3
rm -rf $tmp
