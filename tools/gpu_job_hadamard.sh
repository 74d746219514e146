#!/bin/bash
# the unchanged reference hadamard example with the shim's statistics and a per-call trace of the library
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${TAG:-r04}
for d in "$@"; do
  LSA_SHIM_STATS=1 LSA_TRACE=1 timeout 600 build/reference/hadamard $d > gpurun_out/${TAG}_hadamard_d$d.txt 2> gpurun_out/${TAG}_hadamard_d${d}_trace.txt
  grep micros gpurun_out/${TAG}_hadamard_d$d.txt
  grep lsa_shim_stats gpurun_out/${TAG}_hadamard_d${d}_trace.txt
  grep "\[lsa\] msm" gpurun_out/${TAG}_hadamard_d${d}_trace.txt | awk '{n=$3; sub("n=","",n); t[n]+=$4; c[n]++} END {for (k in t) printf "msm n=%s calls=%d total_ms=%.3f avg_ms=%.3f\n", k, c[k], t[k], t[k]/c[k]}' | sort -t= -k2 -n > gpurun_out/${TAG}_hadamard_d${d}_msm_by_size.txt
  cat gpurun_out/${TAG}_hadamard_d${d}_msm_by_size.txt
done
