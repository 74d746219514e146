#!/bin/bash
# round 5, job T: Fr fold kernels on 29-bit limbs (parity + timing); the bench with its child processes first
cd "$(dirname "$0")/.." || exit 1
timeout 900 python -m pytest tests/test_fr_vec_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -3
python tools/bench_fr_vec.py 2>/dev/null | cut -c1-200
timeout 900 python tools/bench_configs.py --only fr_fold 2>/dev/null | cut -c1-700
timeout 1200 python bench.py > gpurun_out/r05_v3_bench_default.json 2> gpurun_out/r05_v3_bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r05_v3_bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
hp=d['cplink_prover_host_path_ms']; print({k:hp.get(k) for k in ('cold_ms','cold_ms_runs','cold_ms_median','cold_ms_p90','second_ms','warm_ms','cold_error')})
u=d['unchanged_reference_binary']; print(u['timers_ms'], {k:v['ms'] for k,v in u['library_calls'].items()}, u.get('multicore_build',{}).get('timers_ms'))
"
