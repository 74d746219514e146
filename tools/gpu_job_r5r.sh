#!/bin/bash
# round 5, job R: configs after the slot-growth fix; full bench
cd "$(dirname "$0")/.." || exit 1
timeout 900 python tools/bench_configs.py --only g2_msm,cppoly 2>/dev/null | cut -c1-500
timeout 1200 python bench.py > gpurun_out/r05_v2_bench_default.json 2> gpurun_out/r05_v2_bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r05_v2_bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
hp=d['cplink_prover_host_path_ms']
print({k:hp[k] for k in ('cold_ms','cold_ms_runs','cold_ms_median','cold_ms_p90','second_ms','warm_ms')})
u=d['unchanged_reference_binary']; print(u['timers_ms'], u['library_calls'])
for c in d['configs']: print(c['config'][:60], {k:v for k,v in c.items() if k.endswith('ms')})
"
