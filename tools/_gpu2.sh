cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_reference_examples_gpu.py -x -q 2>&1 | tail -8
for d in 8 12 16; do LSA_SEED=3 build/reference/pairing_check $d | tail -1; LSA_SEED=3 LSA_SHIM_EAGER=1 build/reference/pairing_check $d | tail -1; done
LSA_SEED=3 build/reference/hadamard 12 2>&1 | grep -i "micros" | head -12
