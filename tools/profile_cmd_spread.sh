#!/bin/sh
# tools/profile_cmd_spread.sh <tag> <script.py> [args...] -- rocprofv3 kernel trace of `python3 <script.py> args`
# on the GPU box (via gpurun), per-kernel min / median / max into gpurun_out/<tag>_kernel_spread.txt
TAG=$1; shift
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT/prof_$TAG"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG" -o trace -- python3 "$@" > "$OUT/${TAG}_under_rocprof.log" 2>&1
DB=$(find "$OUT/prof_$TAG" -name '*.db' | head -1)
python3 tools/rocpd_summary.py trace "$DB" > "$OUT/${TAG}_kernel_trace_stats.txt"
python3 tools/rocpd_summary.py spread "$DB" > "$OUT/${TAG}_kernel_spread.txt"
rm -rf "$OUT/prof_$TAG"
cat "$OUT/${TAG}_kernel_spread.txt"
