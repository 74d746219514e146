#!/bin/bash
# tools/gpu_job_final.sh -- the round's closing evidence run on the GPU box: the whole GPU suite, the default bench line,
# the rocprofv3 kernel trace of the bench command, the unchanged hadamard 12 / 16 / 20 (d = 20 also under rocprofv3), the
# Fr kernels' timings, and the randomised differential run; everything under gpurun_out/<TAG>_*
cd "$(dirname "$0")/.." || exit 1
TAG=${TAG:-r05_v6}
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > "$OUT/${TAG}_gpu_tests.log"
cat "$OUT/${TAG}_gpu_tests.log"
python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_stderr.txt"
python3 - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','single_call_latency_ms','cplink_prover_ms')}, d['roofline']['frac'])
PY
sh tools/profile_spread.sh $TAG > /dev/null 2>&1
TAG=$TAG sh tools/run_hadamard.sh 12 16 20 > /dev/null 2>&1
grep -h "TOTAL\|lipmaa" "$OUT/${TAG}_hadamard_d20.txt"
python3 tools/bench_fr_vec.py ntt > "$OUT/${TAG}_ntt.txt" 2>&1
python3 tools/bench_fr_vec.py > "$OUT/${TAG}_fr_vec.txt" 2>&1
python3 tools/bench_fr_vec.py sumcheck >> "$OUT/${TAG}_fr_vec.txt" 2>&1
python3 tools/bench_fr_vec.py single >> "$OUT/${TAG}_fr_vec.txt" 2>&1
python3 tools/bench_fr_vec.py prover >> "$OUT/${TAG}_fr_vec.txt" 2>&1
grep -v oracle "$OUT/${TAG}_fr_vec.txt" | tail -8
sh tools/profile_cmd_timeline.sh ${TAG}_fr_vec 12 tools/bench_fr_vec.py > /dev/null 2>&1
python3 tests/fuzz_parity.py ${FUZZ_SECONDS:-300} 5 > "$OUT/${TAG}_fuzz_parity.txt" 2>&1
tail -2 "$OUT/${TAG}_fuzz_parity.txt"
