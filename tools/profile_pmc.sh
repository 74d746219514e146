#!/bin/sh
# tools/profile_pmc.sh <tag> "<counters>" <script.py> [args...] -- run on the GPU box (via gpurun): one rocprofv3
# --pmc pass (counters only: never combined with a trace) of `python3 <script.py> args`, per-kernel counter means
# into gpurun_out/<tag>_pmc.txt
TAG=$1; shift
CTRS=$1; shift
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT/pmc_$TAG"
rocprofv3 --pmc $CTRS -d "$OUT/pmc_$TAG" -o pmc -- python3 "$@" > "$OUT/${TAG}_under_pmc.log" 2>&1
DB=$(find "$OUT/pmc_$TAG" -name '*.db' | head -1)
python3 tools/rocpd_summary.py pmc "$DB" > "$OUT/${TAG}_pmc.txt"
rm -rf "$OUT/pmc_$TAG"
cat "$OUT/${TAG}_pmc.txt"
