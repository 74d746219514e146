#!/bin/bash
# round 5, job K: which bases hadamard's host scalar multiplications use; two more runs of the headline bench
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5k
for d in 12 20; do
LSA_SHIM_BASE_HISTOGRAM=1 LSA_SHIM_STATS=1 build/reference_cmake/src/examples/hadamard $d > gpurun_out/r5k/hadamard$d.out 2> gpurun_out/r5k/hadamard$d.err
grep -h "TOTAL" gpurun_out/r5k/hadamard$d.out; grep -h "histogram" gpurun_out/r5k/hadamard$d.err; grep lsa_shim_stats gpurun_out/r5k/hadamard$d.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['lsa_shim_stats']; print({k: d[k] for k in ('scalar_mul_host','scalar_mul_host_split','msm_g1','msm_g2','pairing','inside_ms')})"
done
for i in 1 2 3; do python bench.py --no-configs --no-cpu-baseline --no-host-path --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f kernel %.4f value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"; done
LSA_NO_LANE_L1=1 python bench.py --no-configs --no-cpu-baseline --no-host-path --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_LANE_L1 ms/step %.4f kernel %.4f value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-host-path --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('200 steps: ms/step %.4f kernel %.4f value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
