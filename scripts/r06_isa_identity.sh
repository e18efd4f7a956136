#!/bin/bash
# Round 6 (no GPU for most of the round): proof that a header change left the DEFAULT kernels of a translation unit untouched -- compile the
# unit from a reference commit and from the working tree for gfx950 (device side only, assembly) and compare the instruction streams.
# Kernel-argument offsets / sizes differ when an argument struct grew; anything else is a real change.
# usage: scripts/r06_isa_identity.sh <commit> <unit.hip>      e.g.  scripts/r06_isa_identity.sh b9d2d09 pqt_rerank_launch.hip
set -e
C=${1:-b9d2d09}; U=${2:-pqt_rerank_launch.hip}
ROOT=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d /tmp/isa_XXXX)
git -C $ROOT archive $C product-quantization-tree_amd/csrc include | tar -x -C $T
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-result -S --cuda-device-only"
hipcc $F -o $T/old.s $T/product-quantization-tree_amd/csrc/$U 2>/dev/null &
hipcc $F -o $T/new.s $ROOT/product-quantization-tree_amd/csrc/$U 2>/dev/null
wait
for v in old new; do sed -e 's/\.Ltmp[0-9]*/.Ltmp/g' $T/$v.s | grep -v '^\s*;' | grep -v '\.ident\|\.file\|\.loc\|__hip_cuid' > $T/$v.n; done
echo "differing lines by kind (numbers masked):"
diff $T/old.n $T/new.n | grep '^[<>]' | sed 's/0x[0-9a-f]*/0xN/g; s/[0-9]\+/N/g' | sort | uniq -c | sort -rn | head -12
echo "differing lines that are not kernel-argument offsets / sizes:"
diff $T/old.n $T/new.n | grep '^[<>]' | grep -v 'offset:\|kernarg\|\.size:\|s_load_dword\|s_add_u32' | head -20
rm -rf $T
