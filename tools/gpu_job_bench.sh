#!/bin/bash
# the default bench line on the GPU box, its wall time, and a digest of the line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${TAG:-r04}
export TMPDIR=/tmp
t0=$(date +%s.%N)
python bench.py "$@" > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
rc=$?
t1=$(date +%s.%N)
echo "bench rc=$rc wall_s=$(python3 -c "print(round($t1 - $t0, 1))")"
tail -5 gpurun_out/${TAG}_bench_default.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print("value %.4g pairs/s  ms_per_step %.4f  single_call %.3f  cplink_prover %.3f" % (d["value"], d["ms_per_step"], d["single_call_latency_ms"], d["cplink_prover_ms"] or 0))
r=d["roofline"]; print("roofline frac %.5f achieved %.1f kernel_ms %.4f traffic %s valu %.3f" % (r["frac"], r["achieved"], r["kernel_ms"], r["traffic"], r["valu"]["frac"]))
print("traffic_detail", json.dumps(r.get("traffic_detail"))[:600])
hp=d.get("cplink_prover_host_path_ms") or {}
print("host path", {k:(round(v,3) if isinstance(v,float) else v) for k,v in hp.items() if k.endswith("_ms") or k=="calls_until_table"})
for c in d.get("configs", []): print({k:(round(v,3) if isinstance(v,float) else v) for k,v in c.items() if k.endswith("ms") or k in ("config","traffic")})
u=d.get("unchanged_reference_binary") or {}
print("reference", u.get("status"), u.get("wall_s"), u.get("timers_ms"), u.get("verifier_check"), {k:v for k,v in (u.get("library_calls") or {}).items() if k in ("msm_g1","pairing")})
print("cpu", d.get("cpu_baseline"))
PY
