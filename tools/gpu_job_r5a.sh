#!/bin/bash
# round 5, job A: the MULTICORE builds on the GPU, the first-sight MSM under the upload settings, hadamard 20 split
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5a
nproc > gpurun_out/r5a/host.txt; cat /sys/kernel/mm/transparent_hugepage/enabled >> gpurun_out/r5a/host.txt
timeout 900 python -m pytest tests/test_reference_examples_gpu.py tests/test_crs_cache_gpu.py tests/test_host_copies_gpu.py -x -q > gpurun_out/r5a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5a/pytest.log
tail -5 gpurun_out/r5a/pytest.log
timeout 900 python tools/cold_msm.py --runs 5 --settings "" --settings "LSA_H2D=direct" --settings "LSA_H2D_THREADS=6" --settings "LSA_H2D_THREADS=12" --settings "LSA_H2D=staged" > gpurun_out/r5a/cold_g1.txt 2>&1
cat gpurun_out/r5a/cold_g1.txt | cut -c1-700
timeout 600 python tools/cold_msm.py --group g2 --runs 3 --settings "" --settings "LSA_H2D=direct" > gpurun_out/r5a/cold_g2.txt 2>&1
cat gpurun_out/r5a/cold_g2.txt | cut -c1-700
for i in 1 2; do
LSA_SHIM_STATS=1 build/reference/hadamard 20 > gpurun_out/r5a/hadamard20_$i.out 2> gpurun_out/r5a/hadamard20_$i.err
grep -h "TOTAL" gpurun_out/r5a/hadamard20_$i.out | head; grep lsa_shim_stats gpurun_out/r5a/hadamard20_$i.err | cut -c1-1200
done
LSA_H2D=direct LSA_SHIM_STATS=1 build/reference/hadamard 20 > gpurun_out/r5a/hadamard20_direct.out 2> gpurun_out/r5a/hadamard20_direct.err
grep -h "TOTAL" gpurun_out/r5a/hadamard20_direct.out | head; grep lsa_shim_stats gpurun_out/r5a/hadamard20_direct.err | cut -c1-1200
