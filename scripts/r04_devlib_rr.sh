#!/bin/bash
# development library for the configs[2]/[3]-shape filter kernels (pqt_rs_query MODE 2): only pqt_rerank_launch.hip is rebuilt, with
# -DPQT_DEV_CFG3_ONLY (LP = 32, C1 = 64, 12 wavefronts, exact filter; everything else of that file reports PQT_ERR_LIMIT), and linked with
# the objects of the last full build in csrc/ -- valid as long as the edit touches nothing the other files compile.
#   usage: bash scripts/r04_devlib_rr.sh <tag> [-DFOO=1 ...]     ->  tune/lib_<tag>.so   (PQT_LIB=... python ...)
set -e
tag=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
src=$root/product-quantization-tree_amd/csrc
obj=$root/tune/obj_$tag
mkdir -p $obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -DPQT_DEV_CFG3_ONLY $@"
(cd $src && hipcc $FLAGS -c -o $obj/pqt_rerank_launch.o pqt_rerank_launch.hip)
hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tune/lib_$tag.so $obj/pqt_rerank_launch.o $src/pqt_hip.o $src/pqt_traverse_launch.o $src/pqt_fused_launch.o $src/pqt_multi.o
ls -la $root/tune/lib_$tag.so
