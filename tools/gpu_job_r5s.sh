#!/bin/bash
# round 5, job S: full GPU suite + the round's profile set on the final tree (TAG r05_v3)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05_v3_gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r05_v3_gpu_tests.log | tail -2
TAG=r05_v3 bash tools/gpu_job_profiles.sh 2>&1 | tail -45
timeout 1200 python bench.py > gpurun_out/r05_v3_bench_default.json 2> gpurun_out/r05_v3_bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r05_v3_bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
hp=d['cplink_prover_host_path_ms']; print({k:hp[k] for k in ('cold_ms','cold_ms_runs','cold_ms_median','cold_ms_p90','second_ms','warm_ms')})
for c in d['configs']: print(c['config'][:60], {k:v for k,v in c.items() if k.endswith('ms')})
"
