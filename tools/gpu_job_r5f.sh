#!/bin/bash
# round 5, job F: full GPU suite after the round's changes so far; cold G2 with k_prepare_g2; hadamard 12/16/20
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5f
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5f/pytest_gpu.log 2>&1; tail -5 gpurun_out/r5f/pytest_gpu.log
timeout 600 python tools/cold_msm.py --group g2 --runs 3 --settings "" --settings LSA_G2_PREPARE_OLD=1 > gpurun_out/r5f/cold_g2.txt 2>&1
timeout 600 python tools/cold_msm.py --group g1 --runs 5 --settings "" > gpurun_out/r5f/cold_g1.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5f/cold_*.txt')):
    print(f)
    for l in open(f):
        if l.startswith('{"cold_msm"'):
            d=json.loads(l)['cold_msm']
            sp=d['cold_split']
            med=lambda k: sorted(x[k] for x in sp)[len(sp)//2] if sp else None
            print(' ', d['setting'], 'cold', d['cold_ms_runs'], 'h2d', med('h2d_scalars_ms'), 'prep', med('bases_prepare_ms'), 'msm', med('msm_ms'), 'second', d['second_ms'], 'third', d['third_ms'], 'ok', d['all_ok'], d['errors'][:1])
        elif l.strip(): print('  ??', l[:300])
PY
for d in 12 16 20; do
for v in "" "LSA_NO_COMPACT_G2=1"; do
env $v LSA_SHIM_STATS=1 build/reference/hadamard $d > gpurun_out/r5f/hadamard${d}_$v.out 2> gpurun_out/r5f/hadamard${d}_$v.err
echo "hadamard $d $v"; grep -h "TOTAL" gpurun_out/r5f/hadamard${d}_$v.out | head; grep lsa_shim_stats gpurun_out/r5f/hadamard${d}_$v.err | cut -c1-900
done; done
LSA_TRACE=1 build/reference/hadamard 20 > /dev/null 2> gpurun_out/r5f/hadamard20_trace.err
grep "msm " gpurun_out/r5f/hadamard20_trace.err | awk '$3 ~ /n=(104|209|52)/' | head -40
