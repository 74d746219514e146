#!/bin/sh
# tools/profile_cmd.sh <tag> <script.py> [args...] -- run on the GPU box (via gpurun): rocprofv3
# kernel trace of `python3 <script.py> args`, summarised into gpurun_out/<tag>_kernel_trace_stats.txt
TAG=$1; shift
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT/prof_$TAG"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG" -o trace -- python3 "$@" > "$OUT/${TAG}_under_rocprof.log" 2>&1
DB=$(find "$OUT/prof_$TAG" -name '*.db' | head -1)
python3 tools/rocpd_summary.py trace "$DB" > "$OUT/${TAG}_kernel_trace_stats.txt"
rm -rf "$OUT/prof_$TAG"
head -30 "$OUT/${TAG}_kernel_trace_stats.txt"
