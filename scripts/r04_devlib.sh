#!/bin/bash
# development library for kernel experiments: tune/lib_<tag>.so built from the CURRENT sources with -DPQT_DEV_SIFT1M_ONLY (only the
# kernels the SIFT1M-shape headline launches: ~40 s instead of ~3 min) plus any extra -D flags.  Run benches against it with
#   PQT_LIB=$PWD/tune/lib_<tag>.so python bench.py ...      (scripts/tune_libs.sh runs all tune/lib_*.so)
#   usage: bash scripts/r04_devlib.sh <tag> [-DFOO=1 ...]
set -e
tag=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
src=$root/product-quantization-tree_amd/csrc
obj=$root/tune/obj_$tag
mkdir -p $obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -DPQT_DEV_SIFT1M_ONLY $@"
for f in pqt_hip pqt_rerank_launch pqt_traverse_launch pqt_fused_launch; do
  (cd $src && hipcc $FLAGS -c -o $obj/$f.o $f.hip) &
done
(cd $src && hipcc -O2 -std=c++17 -fPIC -Wall -Wno-unused-result -c -o $obj/pqt_multi.o pqt_multi.cpp) &
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tune/lib_$tag.so $obj/*.o
ls -la $root/tune/lib_$tag.so
