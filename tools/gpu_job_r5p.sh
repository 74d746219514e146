#!/bin/bash
# round 5, job P: first calls with the workspaces sized at init
cd "$(dirname "$0")/.." || exit 1
LSA_TRACE=2 build/reference_cmake/src/examples/hadamard 20 2>&1 >/dev/null | grep -E "grow|msm_g2|msm_g1 +n=(1048576|524288)" | head -24
run() { python tools/cold_msm.py "$@" 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"cold_msm\"'):
        d=json.loads(l)['cold_msm']; sp=d['cold_split']
        med=lambda k: sorted(x[k] for x in sp)[len(sp)//2] if sp else None
        print(' ', d['group'], d['setting'], 'cold', d['cold_ms_runs'], '| prep', med('bases_prepare_ms'), 'msm', med('msm_ms'), 'second', d['second_ms'][:3], d['errors'][:1])
    elif l.strip(): print('??', l[:200])
"; }
run --runs 6 --settings ""
run --runs 5 --group g2 --settings ""
for i in 1 2 3; do LSA_SHIM_STATS=1 build/reference_cmake/src/examples/hadamard 20 2>&1 >/dev/null | grep lsa_shim_stats | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['lsa_shim_stats']; print({k: d[k]['ms'] for k in ('msm_g1','msm_g2','batch_exp','pairing','scalar_mul_host')}, d['msm_host_path'])"; done
python -c "
import time,sys
sys.path.insert(0,'.')
t=time.perf_counter()
import legosnark_amd as lsa
t1=time.perf_counter(); lsa.init(0); print('import %.0f ms, lsa_init %.0f ms' % ((t1-t)*1e3, (time.perf_counter()-t1)*1e3))"
