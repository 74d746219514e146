cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pairing_precomp_gpu.py tests/test_pairing_gpu.py tests/test_public_vectors.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -5
timeout 600 python tools/pairing_shapes.py 2>&1 | tail -30
