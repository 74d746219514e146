#!/bin/bash
# round 5, job M: full GPU suite on the current tree, then the round's profile set
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05_v1_gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r05_v1_gpu_tests.log | tail -2
TAG=r05_v1 bash tools/gpu_job_profiles.sh 2>&1 | tail -40
timeout 1200 python bench.py > gpurun_out/r05_v1_bench_default.json 2> gpurun_out/r05_v1_bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r05_v1_bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
"
