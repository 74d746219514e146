#!/bin/bash
# headline step time against the number of tail slots in use, alternating, same box
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2 3; do
  for s in 4 8 6; do
    LSA_TAIL_SLOTS=$s python bench.py --no-pmc --no-configs --no-host-path --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('slots $s', round(d['value']/1e8,3), round(d['ms_per_step'],4))"
  done
done
