cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r03_bench_a.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_bench_a.json"))
print({k:d[k] for k in ("value","ms_per_step","single_call_latency_ms","cplink_prover_ms")})
print(d["roofline"]["frac"], d["roofline"]["valu"]["frac"], d["roofline"]["traffic"])
hp=d["cplink_prover_host_path_ms"]; print({k:hp[k] for k in ("cold_ms","second_ms","warm_ms")})
for c in d["configs"]: print(json.dumps(c)[:400])
print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("multicore",{}).get("value"))
PY
