cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03_v2_bench_default.json 2> gpurun_out/r03_v2_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_v2_bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","single_call_latency_ms","cplink_prover_ms")})
print(d["roofline"]["frac"], d["roofline"]["valu"]["frac"], d["roofline"]["traffic"])
hp=d["cplink_prover_host_path_ms"]; print({k:hp[k] for k in ("cold_ms","second_ms","ms_while_the_copies_are_built","calls_until_table","warm_ms")})
for c in d["configs"]: print(c["config"][:90], c.get("ms", c.get("commit_ms")), c.get("valu",{}).get("frac"))
PY
