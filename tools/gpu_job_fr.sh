#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
python3 tools/bench_fr_vec.py ntt > gpurun_out/r05_v4_ntt.txt 2>&1
sh tools/profile_cmd_timeline.sh r05_v4_ntt 12 tools/bench_fr_vec.py ntt > /dev/null 2>&1
python3 tools/bench_fr_vec.py > gpurun_out/r05_v4_fr_vec.txt 2>&1
sh tools/profile_cmd_timeline.sh r05_v4_fr_vec 40 tools/bench_fr_vec.py > /dev/null 2>&1
grep -E '"log_n": (20|24)' gpurun_out/r05_v4_ntt.txt | cut -c1-160; cat gpurun_out/r05_v4_ntt_kernel_spread.txt; grep -E "k_fold|kernel " gpurun_out/r05_v4_fr_vec_kernel_spread.txt
