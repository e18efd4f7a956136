#!/bin/bash
# VGPRs / spilled VGPRs / scratch bytes of every kernel one .hip file compiles to (device-only compile, code-object notes)
#   usage: bash scripts/kernel_resources.sh pqt_rerank_launch.hip [-DPQT_DEV_CFG3_ONLY ...]
set -e
root=$(cd $(dirname $0)/.. && pwd)
f=$1; shift
tmp=$(mktemp -d)
(cd $root/product-quantization-tree_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" --cuda-device-only -c -o $tmp/dev.bundle $f)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/dev.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.elf > $tmp/notes.txt
python3 - $tmp/notes.txt <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
print("vgpr spilled scratch_bytes sgpr lds_static kernel")
for k in re.split(r'\n\s+- \.agpr_count', t)[1:]:
    g = lambda key: int(re.search(r'\.%s:\s+(\d+)' % key, k).group(1))
    name = subprocess.run(['c++filt', re.search(r'\.name:\s+(\S+)', k).group(1)], capture_output=True, text=True).stdout.strip()
    print(g('vgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('sgpr_count'), g('group_segment_fixed_size'), name[:160])
PY
rm -rf $tmp
