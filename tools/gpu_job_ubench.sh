#!/bin/bash
# tools/gpu_job_ubench.sh -- the two issue-rate microbenchmarks (profiles/r04_ubench_*.txt)
set -u
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o /tmp/ubench_issue 2>/dev/null && /tmp/ubench_issue > gpurun_out/r04_ubench_issue_rates.txt 2>&1
hipcc -O3 --offload-arch=gfx950 -std=c++17 -I legosnark_amd/csrc tools/ubench_int.hip -o /tmp/ubench_int 2>/dev/null && /tmp/ubench_int > gpurun_out/r04_ubench_field_mul.txt 2>&1
tail -30 gpurun_out/r04_ubench_issue_rates.txt
