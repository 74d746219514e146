cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_crs_cache_gpu.py tests/test_msm_gpu.py -m gpu -x -q 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -8
for N in 4096 65536 262144; do build/h2d_vectors x $N | grep "rep 2\|n=$N "; done
export LSA_SHIM_STATS=1
for d in 12 16; do build/reference/hadamard $d 2>&1 | grep "Prove\|msm_g1" | sed "s/.*\"msm_g1\"/msm_g1/" | cut -c1-120; done
