#!/bin/sh
# tools/run_hadamard.sh [d ...] -- on the GPU box (via gpurun): the reference's UNCHANGED hadamard example (built against the
# shim by __graft_entry__.build()) with ITS OWN timers (##had_sc ... Prove/Verify: N micros, src/examples/hadamard.cc:98-105,
# src/utils/benchmark.cc:3-6) at the given numbers of variables, into gpurun_out/<tag>_hadamard_d<d>.txt; for the last d also
# under rocprofv3 --kernel-trace (program directly after --) with the per-kernel summary next to it.
TAG=${TAG:-r03}
# the binary the reference's own CMakeLists.txt builds (its flags, -O3 -ggdb) when it is there, else the Makefile build
HAD=build/reference_cmake/src/examples/hadamard
[ -x "$HAD" ] || HAD=build/reference/hadamard
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
last=""
for d in "$@"; do
  t0=$(date +%s.%N)
  timeout 1500 env LSA_SHIM_STATS=1 $HAD $d > "$OUT/${TAG}_hadamard_d$d.txt" 2>&1
  rc=$?
  t1=$(date +%s.%N)
  echo "d=$d rc=$rc wall_s=$(echo "$t1 - $t0" | bc 2>/dev/null || python3 -c "print($t1 - $t0)")" | tee -a "$OUT/${TAG}_hadamard_d$d.txt"
  grep "micros" "$OUT/${TAG}_hadamard_d$d.txt"
  last=$d
done
if [ -n "$last" ]; then
  mkdir -p "$OUT/prof_had"
  timeout 1200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_had" -o trace -- $HAD $last > "$OUT/${TAG}_hadamard_d${last}_under_rocprof.txt" 2>&1
  DB=$(find "$OUT/prof_had" -name '*.db' | head -1)
  python3 tools/rocpd_summary.py trace "$DB" > "$OUT/${TAG}_hadamard_d${last}_kernel_trace_stats.txt"
  python3 tools/rocpd_summary.py spread "$DB" > "$OUT/${TAG}_hadamard_d${last}_kernel_spread.txt"
  rm -rf "$OUT/prof_had"
  head -25 "$OUT/${TAG}_hadamard_d${last}_kernel_spread.txt"
fi
