#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 900 python -m pytest tests/test_fr_vec_gpu.py -x -q 2>&1 | tail -2
python tools/bench_fr_vec.py 2>/dev/null | cut -c1-200
sh tools/profile_cmd_timeline.sh r5u_fold 60 tools/bench_fr_vec.py > /dev/null 2>&1; grep -E "k_fold|kernel " gpurun_out/r5u_fold_kernel_spread.txt; tail -30 gpurun_out/r5u_fold_timeline.txt
