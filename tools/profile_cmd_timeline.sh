#!/bin/sh
# tools/profile_cmd_timeline.sh <tag> <last-kernels> <script.py> [args...] -- rocprofv3 kernel trace of
# `python3 <script.py> args` on the GPU box; per-kernel spread and the timeline of the last N dispatches
TAG=$1; shift
LAST=$1; shift
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT/prof_$TAG"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG" -o trace -- python3 "$@" > "$OUT/${TAG}_under_rocprof.log" 2>&1
DB=$(find "$OUT/prof_$TAG" -name '*.db' | head -1)
python3 tools/rocpd_summary.py spread "$DB" > "$OUT/${TAG}_kernel_spread.txt"
python3 tools/rocpd_summary.py timeline "$DB" "$LAST" > "$OUT/${TAG}_timeline.txt"
rm -rf "$OUT/prof_$TAG"
cat "$OUT/${TAG}_kernel_spread.txt"; cat "$OUT/${TAG}_timeline.txt"; tail -3 "$OUT/${TAG}_under_rocprof.log"
