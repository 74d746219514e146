#!/bin/bash
# round 5, job H: full GPU suite after the NTT v2 / tail-ordering changes, NTT timing, the headline bench
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5h
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5h/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r5h/pytest_gpu.log | tail -3
timeout 600 python tools/bench_fr_vec.py ntt > gpurun_out/r5h/ntt_bench.txt 2>&1; cat gpurun_out/r5h/ntt_bench.txt | cut -c1-200
timeout 900 python bench.py > gpurun_out/r5h/bench.json 2> gpurun_out/r5h/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5h/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'lat', d.get('single_call_latency_ms'), 'cplink', d.get('cplink_prover_ms'))
print(json.dumps(d.get('cplink_prover_host_path_ms'))[:800])
for c in d.get('configs', []): print(json.dumps(c)[:400])
PY
