#!/bin/bash
# round 5, job L: per-call split of hadamard 20's G2 MSMs (box-to-box spread), then the GPU suite's reference tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5l
for i in 1 2; do
LSA_TRACE=1 LSA_SHIM_STATS=1 build/reference_cmake/src/examples/hadamard 20 > gpurun_out/r5l/hadamard20_$i.out 2> gpurun_out/r5l/hadamard20_$i.err
grep -h "TOTAL" gpurun_out/r5l/hadamard20_$i.out
grep -A1 "msm_g2" gpurun_out/r5l/hadamard20_$i.err | grep -v "^--" | paste - - | awk '{print $2, $3, $4, $8, $9, $10, $11, $12, $13}'
grep lsa_shim_stats gpurun_out/r5l/hadamard20_$i.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['lsa_shim_stats']; print({k: d[k] for k in ('msm_g1','msm_g2','msm_host_path')})"
done
timeout 900 python -m pytest tests/test_reference_examples_gpu.py -x -q 2>&1 | tail -3
