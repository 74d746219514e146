#!/bin/sh
# tools/profile_bench.sh <tag> [bench args...] -- run on the GPU box (via gpurun): rocprofv3 kernel
# trace of bench.py, summarised into gpurun_out/<tag>_kernel_trace_stats.txt
TAG=$1; shift
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT/prof_$TAG"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG" -o trace -- python3 bench.py --no-cpu-baseline --no-host-path --no-pmc --no-configs "$@" > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_bench_under_rocprof.log"
DB=$(find "$OUT/prof_$TAG" -name '*.db' | head -1)
python3 tools/rocpd_summary.py trace "$DB" > "$OUT/${TAG}_kernel_trace_stats.txt"
rm -rf "$OUT/prof_$TAG"
head -40 "$OUT/${TAG}_kernel_trace_stats.txt"
