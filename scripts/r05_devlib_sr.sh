#!/bin/bash
# development library for the shared-row pass: only pqt_shared_launch.hip is rebuilt (seconds) and linked with the objects of the last full
# build in csrc/ -- valid as long as the edit touches nothing the other translation units compile.
#   usage: bash scripts/r05_devlib_sr.sh <tag> [-DPQT_SR_TILE=1024u -DPQT_SR_U=2 ...]   ->  tune/lib_<tag>.so   (PQT_LIB=... python ...)
set -e
tag=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
src=$root/product-quantization-tree_amd/csrc
obj=$root/tune/obj_$tag
mkdir -p $obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result $@"
(cd $src && hipcc $FLAGS -c -o $obj/pqt_shared_launch.o pqt_shared_launch.hip)
hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tune/lib_$tag.so $obj/pqt_shared_launch.o $src/pqt_hip.o $src/pqt_rerank_launch.o $src/pqt_traverse_launch.o $src/pqt_fused_launch.o $src/pqt_multi.o
ls -la $root/tune/lib_$tag.so
