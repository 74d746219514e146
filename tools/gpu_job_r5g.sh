#!/bin/bash
# round 5, job G: the three-pass NTT: parity, timing, kernel profile
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5g
timeout 1200 python -m pytest tests/test_fr_vec_gpu.py tests/test_host_copies_gpu.py tests/test_shim_io.py -x -q 2>&1 | tail -15
timeout 600 python tools/bench_fr_vec.py ntt > gpurun_out/r5g/ntt_bench.txt 2>&1; cat gpurun_out/r5g/ntt_bench.txt
sh tools/profile_cmd_timeline.sh r5g_ntt 12 tools/bench_fr_vec.py ntt > gpurun_out/r5g/ntt_profile.txt 2>&1
grep -E "k_ntt|kernel " gpurun_out/r5g_ntt_kernel_spread.txt
LSA_SHIM_STATS=1 LSA_TRACE=1 build/reference/hadamard 20 > gpurun_out/r5g/hadamard20.out 2> gpurun_out/r5g/hadamard20.err
grep -h "TOTAL\|lipmaa" gpurun_out/r5g/hadamard20.out | head; grep fr_ntt gpurun_out/r5g/hadamard20.err | head -8
