#!/bin/bash
# round 5, job E: one copy stream; the G2 first-call cost (scratch); G2 compact pipeline parity and G2-vs-n
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5e
timeout 900 python tools/cold_msm.py --runs 8 --settings "" > gpurun_out/r5e/cold_g1.txt 2>&1
timeout 900 python tools/cold_msm.py --group g2 --runs 3 --settings "" --settings HSA_NO_SCRATCH_RECLAIM=1 --settings HSA_SCRATCH_SINGLE_LIMIT=4000000000 > gpurun_out/r5e/cold_g2.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5e/cold_*.txt')):
    print(f)
    for l in open(f):
        if l.startswith('{"cold_msm"'):
            d=json.loads(l)['cold_msm']
            sp=d['cold_split']
            med=lambda k: sorted(x[k] for x in sp)[len(sp)//2] if sp else None
            print(' ', d['setting'], 'cold', d['cold_ms_runs'], 'h2d', med('h2d_scalars_ms'), 'prep', med('bases_prepare_ms'), 'msm', med('msm_ms'), 'second', d['second_ms'], 'third', d['third_ms'], 'ok', d['all_ok'], d['errors'][:1])
        elif l.strip(): print('  ??', l[:300])
PY
sh tools/profile_cmd_timeline.sh r5e_g2cold 80 tools/cold_msm.py --child --group g2 > gpurun_out/r5e/g2cold_profile.txt 2>&1
tail -85 gpurun_out/r5e/g2cold_profile.txt | cut -c1-160
timeout 1500 python -m pytest tests/test_msm_compact_g2_gpu.py tests/test_msm_compact_gpu.py -x -q 2>&1 | tail -15
python tools/msm_vs_n.py g2 16 16 > gpurun_out/r5e/g2_vs_n_compact.txt 2>&1; cat gpurun_out/r5e/g2_vs_n_compact.txt
LSA_NO_COMPACT_G2=1 python tools/msm_vs_n.py g2 16 16 > gpurun_out/r5e/g2_vs_n_general.txt 2>&1; cat gpurun_out/r5e/g2_vs_n_general.txt
