#!/bin/bash
# round 5, job I: the fused Miller kernel's ceiling (timing-only experiments), NTT with the unpacked W table, new bench configs
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5i
for e in 0 1 2 4 3 5 6 7; do LSA_FUSED_EXPERIMENT=$e python tools/fused_experiment.py 2>/dev/null | tail -1; done | tee gpurun_out/r5i/fused_experiment.txt
timeout 600 python tools/bench_fr_vec.py ntt 2>/dev/null | grep -E '"log_n": (20|22|24)' | cut -c1-200 | tee gpurun_out/r5i/ntt_bench.txt
timeout 900 python tools/bench_configs.py --only fr_fold,ntt 2>/dev/null | tee gpurun_out/r5i/configs_fr.txt | cut -c1-900
timeout 600 python -m pytest tests/test_fr_vec_gpu.py -x -q 2>&1 | tail -2
