cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pairing_precomp_gpu.py tests/test_pairing_gpu.py tests/test_public_vectors.py tests/test_reference_examples_gpu.py -m gpu -x -q 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -5
export LSA_SHIM_STATS=1
for d in 12 20; do build/reference/hadamard $d 2>&1 | grep "Verify\|pairing" | sed "s/.*\"pairing\"/pairing/" | cut -c1-220; done
build/reference/pairing_check 20 | tail -2
