#!/bin/bash
# A/B of one environment switch on the headline step: tools/gpu_job_ab.sh VAR [steps] [reps]
cd "$(dirname "$0")/.." || exit 1
VAR=$1; STEPS=${2:-200}; REPS=${3:-3}
one() { python bench.py --no-configs --steps $STEPS --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step  kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for r in $(seq $REPS); do
  echo -n "default      : "; one
  echo -n "$VAR=1 : "; env $VAR=1 python bench.py --no-configs --steps $STEPS --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step  kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
