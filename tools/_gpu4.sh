cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_crs_cache_gpu.py tests/test_msm_gpu.py tests/test_reference_examples_gpu.py tests/test_comm.py -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py --no-pmc --no-configs --no-cpu-baseline > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r03_bench_b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_bench_b.json"))
print({k:d[k] for k in ("value","ms_per_step","single_call_latency_ms","cplink_prover_ms")})
hp=d["cplink_prover_host_path_ms"]; print({k:hp[k] for k in hp if k not in ("cold","second","warm","note","call")})
print(hp["second"]); print(hp["warm"])
PY
