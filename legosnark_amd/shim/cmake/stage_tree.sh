#!/bin/sh
# stage_tree.sh <legosnark checkout> <out dir>
# Builds a CMake source tree for an UNTOUCHED (possibly read-only) LegoSNARK checkout whose
# `depends/libsnark` and `depends/fmt` submodules are replaced by the packages in this directory:
# symlinks only, nothing is copied.  Then:
#   cmake -S <out dir> -B <build dir> -DWITH_PROCPS=OFF && cmake --build <build dir> --target cplink hadamard matrixsc
# (legogrothmatrix needs all of libsnark's R1CS / Groth16 and is out of scope.)
set -e
REF=$(cd "$1" && pwd)
OUT=$2
HERE=$(cd "$(dirname "$0")" && pwd)
rm -rf "$OUT"
mkdir -p "$OUT/depends"
ln -s "$REF/CMakeLists.txt" "$OUT/CMakeLists.txt"
ln -s "$REF/src" "$OUT/src"
ln -s "$REF/depends/CMakeLists.txt" "$OUT/depends/CMakeLists.txt"
ln -s "$HERE/libsnark" "$OUT/depends/libsnark"
ln -s "$HERE/fmt" "$OUT/depends/fmt"
echo "staged $OUT"
