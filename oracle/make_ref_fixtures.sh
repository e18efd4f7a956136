#!/bin/bash
# oracle/make_ref_fixtures.sh -- the oracle PIN recipe.  Run on a machine that has
#   * the reference checkout (REF, default /root/reference) and
#   * a real Eigen 3 (EIGEN_DIR = the directory that contains Eigen/Dense; the reference's CMake points at ~/libs/eigen,
#     cpu_version/CMakeLists.txt:5,22).
# It compiles the UNMODIFIED cpu_version headers with the reference's own flags (cpu_version/CMakeLists.txt:20: -O3 -Ofast,
# -std=c++11), runs the reference's loadTree -> insert -> saveBins -> query on the inputs of tests/golden/dump_small.*, and
# overwrites tests/golden/dump_small.bins + dump_small_expected.npz with what the REFERENCE produced.  After that
#   python -m pytest tests/test_cpu_oracle.py tests/test_gpu_tools.py -k dump
# compares the restatement (CPU) and the HIP engine (GPU) with a genuine run of the reference: "parity unpinned" for the
# traversal becomes pinned.  Nothing here runs in the build image (no Eigen, no network) and no stand-in for
# Eigen is provided: without EIGEN_DIR the script stops.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
REF="${REF:-/root/reference}"
: "${EIGEN_DIR:?set EIGEN_DIR to the directory that contains Eigen/Dense (a real Eigen 3 checkout)}"
[ -f "$EIGEN_DIR/Eigen/Dense" ] || { echo "no Eigen/Dense under $EIGEN_DIR" >&2; exit 1; }
[ -f "$REF/cpu_version/quantizer/treequantizer.hpp" ] || { echo "no reference under $REF" >&2; exit 1; }
mkdir -p "$HERE/_ref"
TMP="$(mktemp -d)"
# 1. the inputs of the committed fixture (same seeds): tree dump written by the restatement in the reference's .tree format,
#    raw base / query vectors
python3 "$ROOT/tests/golden/make_dump_fixture.py" --inputs-only "$TMP"
# 2. the genuine reference, its own flags
g++ -std=c++11 -O3 -Ofast -w -I"$EIGEN_DIR" -I"$REF/cpu_version" -o "$HERE/_ref/ref_cpu_driver" "$HERE/ref_cpu_driver.cpp"
# 3. run it
"$HERE/_ref/ref_cpu_driver" "$TMP/dump_small.tree" "$TMP/base.raw" "$(cat "$TMP/n")" "$TMP/queries.raw" "$(cat "$TMP/nq")" 1500 400 "$TMP/ref.bins" "$TMP/ref.lists"
# 4. install as the golden fixture (pinned_by = "reference")
python3 "$ROOT/tests/golden/make_dump_fixture.py" --install-reference "$TMP"
echo "tests/golden/dump_small.* now hold a genuine reference run; re-run the dump tests."
