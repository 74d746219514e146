#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for s in "" "LSA_WARM_MB=192" "LSA_WARM=0"; do python tools/cold_msm.py --runs 4 --settings "$s" 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"cold_msm\"'):
        d=json.loads(l)['cold_msm']
        print(d['setting'], 'init', d['import_and_lsa_init_ms'], 'cold', d['cold_ms_runs'])
"; done
