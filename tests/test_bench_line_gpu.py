"""GPU: `python bench.py` (short: 3 steps, no PMC child runs, no CPU baseline) prints ONE JSON line with the contract's
keys -- the BASELINE.json metric on config 2, the roofline object of the dominant kernel, the result-checked `configs`
block and the unchanged reference binary's own timers -- and every time in it belongs to a checked result."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_shape():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-pmc", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "configs", "cplink_prover_host_path_ms", "unchanged_reference_binary"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "pairs/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "2^20" in d["metric"] and "workload" in d["config"] and d["value"] > 1e8
    assert abs(d["value"] - (1 << 20) / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["traffic"] is None   # --no-pmc
    assert rf["algorithmic_bytes_per_launch"] == 96 << 20 and 0.3 < rf["valu"]["frac"] < 1.0
    assert d["result_checked_by_identity"] is True and d["cplink_prover_host_path_ms"]["all_results_checked"] is True
    assert len(d["configs"]) >= 5 and all(c.get("result_checked") for c in d["configs"])
    ref = d["unchanged_reference_binary"]
    if ref["status"].startswith("rc="):                       # the binary travels with the snapshot when it was built
        assert ref["status"] == "rc=0" and ref["timers_ms"]["had_sc TOTAL Prove"] > 0 and ref["timers_ms"]["had_sc TOTAL Verify"] > 0
        assert 0 < ref["inside_library_ms"] < ref["process_ms"]
