#!/bin/sh
# tools/profile_spread.sh <tag> [bench args...] -- like profile_bench.sh, but prints per-kernel
# min / median / max durations (the mean hides warm-up launches and the other workloads bench.py runs)
TAG=$1; shift
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT/prof_$TAG"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG" -o trace -- python3 bench.py --no-cpu-baseline --no-host-path --no-pmc --no-configs "$@" > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_bench_under_rocprof.log"
DB=$(find "$OUT/prof_$TAG" -name '*.db' | head -1)
python3 tools/rocpd_summary.py trace "$DB" > "$OUT/${TAG}_kernel_trace_stats.txt"
python3 tools/rocpd_summary.py spread "$DB" > "$OUT/${TAG}_kernel_spread.txt"
rm -rf "$OUT/prof_$TAG"
cat "$OUT/${TAG}_kernel_spread.txt"
