#!/bin/bash
# round 5, job J: after the pruning: pairing tests, the fused kernel's split with Z = 1 inputs, a whole bench.py run
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5j
timeout 1500 python -m pytest tests/test_pairing_gpu.py tests/test_pairing_precomp_gpu.py tests/test_g2_pair_kernel_gpu.py tests/test_msm_gpu.py -x -q 2>&1 | tail -4
for e in 0 1 2 4 3 5 6 7; do LSA_FUSED_EXPERIMENT=$e python tools/fused_experiment.py 2>/dev/null | tail -1; done | tee gpurun_out/r5j/fused_experiment.txt
timeout 1200 python bench.py > gpurun_out/r5j/bench.json 2> gpurun_out/r5j/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5j/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'lat', d.get('single_call_latency_ms'), 'cplink', d.get('cplink_prover_ms'))
hp=d.get('cplink_prover_host_path_ms') or {}
print({k: hp.get(k) for k in ('cold_ms','cold_ms_runs','cold_ms_median','cold_ms_p90','second_ms','warm_ms','cold_error','transparent_hugepage')})
print(json.dumps(d.get('unchanged_reference_binary'))[:1800])
for c in d.get('configs', []): print(json.dumps(c)[:300])
PY
