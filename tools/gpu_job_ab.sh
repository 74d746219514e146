#!/bin/bash
# A/B of an environment switch on the headline bench line, alternating on one box:  tools/gpu_job_ab.sh VAR=val [reps]
cd "$(dirname "$0")/.." || exit 1
SW=$1; REPS=${2:-3}
for rep in $(seq 1 $REPS); do
  for mode in default "$SW"; do
    if [ "$mode" = default ]; then E=""; else E="$SW"; fi
    env $E python bench.py --no-pmc --no-configs --no-host-path --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', round(d['value']/1e8,3), 'ms/step', round(d['ms_per_step'],4), 'single', round(d['single_call_latency_ms'],3), 'cplink', round(d['cplink_prover_ms'],3))"
  done
done
