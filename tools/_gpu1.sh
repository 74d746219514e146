set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pairing_precomp_gpu.py tests/test_pairing_gpu.py -x -q 2>&1 | tail -15
timeout 600 python tools/bench_configs.py --only pairing,cphad_verify 2>&1 | tail -8
