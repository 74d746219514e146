#!/bin/bash
# round 5, job B: first-sight MSM: copy streams, huge vs small host pages
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5b
S='--settings "" --settings LSA_H2D=direct --settings LSA_H2D_STREAMS=0 --settings LSA_H2D_STREAMS=2 --settings LSA_H2D_THREADS=6'
eval timeout 900 python tools/cold_msm.py --runs 5 $S > gpurun_out/r5b/cold_g1.txt 2>&1
eval timeout 900 python tools/cold_msm.py --runs 5 --small-pages $S > gpurun_out/r5b/cold_g1_small.txt 2>&1
timeout 600 python tools/cold_msm.py --group g2 --runs 3 --settings "" --settings "LSA_H2D=direct" > gpurun_out/r5b/cold_g2.txt 2>&1
timeout 600 python tools/cold_msm.py --group g2 --runs 3 --small-pages --settings "" --settings "LSA_H2D=direct" > gpurun_out/r5b/cold_g2_small.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5b/cold_*.txt')):
    print(f)
    for l in open(f):
        if l.startswith('{"cold_msm"'):
            d=json.loads(l)['cold_msm']
            sp=d['cold_split']
            med=lambda k: sorted(x[k] for x in sp)[len(sp)//2] if sp else None
            print(' ', d['setting'], 'cold', d['cold_ms_runs'], 'h2d', med('h2d_scalars_ms'), 'prep', med('bases_prepare_ms'), 'msm', med('msm_ms'), 'second', d['second_ms'], 'third', d['third_ms'], 'ok', d['all_ok'], d['errors'][:1])
        elif l.strip(): print('  ??', l[:300])
PY
for i in 1 2; do
LSA_SHIM_STATS=1 build/reference/hadamard 20 > gpurun_out/r5b/hadamard20_$i.out 2> gpurun_out/r5b/hadamard20_$i.err
grep -h "TOTAL" gpurun_out/r5b/hadamard20_$i.out | head; grep lsa_shim_stats gpurun_out/r5b/hadamard20_$i.err | cut -c1-1200
done
