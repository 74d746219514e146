#!/bin/bash
# A/B of one environment switch on the non-headline configs: tools/gpu_job_ab_configs.sh VAR configs [reps]
cd "$(dirname "$0")/.." || exit 1
VAR=$1; CFG=$2; REPS=${3:-3}
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' | '.join(' '.join('%s=%.3f' % (k, v) for k, v in c.items() if k.endswith('ms')) for c in d.get('configs', [])))"; }
for r in $(seq $REPS); do
  echo -n "default : "; python bench.py --only-configs $CFG 2>/dev/null | show
  echo -n "$VAR=1 : "; env $VAR=1 python bench.py --only-configs $CFG 2>/dev/null | show
done
