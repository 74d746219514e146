#!/bin/bash
# round 5, job N: the 15-ms bubble in front of a first MSM: staged large downloads, temporaries freed or kept
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5n
run() { python tools/cold_msm.py "$@" 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"cold_msm\"'):
        d=json.loads(l)['cold_msm']; sp=d['cold_split']
        med=lambda k: sorted(x[k] for x in sp)[len(sp)//2] if sp else None
        print(' ', d['setting'], 'small', d['small_pages'], 'free', d.get('free_temporaries'), 'cold', d['cold_ms_runs'], '| prep', med('bases_prepare_ms'), 'msm', med('msm_ms'), 'second', d['second_ms'][:3], d['errors'][:1])
    elif l.strip(): print('??', l[:200])
"; }
echo "g1"; run --runs 6 --settings "" --settings LSA_H2D=direct
echo "g1 free temporaries"; run --runs 6 --free-temporaries --settings "" --settings LSA_H2D=direct
echo "g1 small pages"; run --runs 6 --small-pages --settings "" --settings LSA_H2D=direct
echo "g1 small pages, free temporaries"; run --runs 6 --small-pages --free-temporaries --settings ""
echo "g2"; run --runs 4 --group g2 --settings "" --settings LSA_H2D=direct
echo "g2 free temporaries"; run --runs 4 --group g2 --free-temporaries --settings ""
for i in 1 2; do LSA_SHIM_STATS=1 build/reference_cmake/src/examples/hadamard 20 2>&1 >/dev/null | grep lsa_shim_stats | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['lsa_shim_stats']; print({k: d[k]['ms'] for k in ('msm_g1','msm_g2','batch_exp','pairing','scalar_mul_host')}, d['msm_host_path'])"; done
