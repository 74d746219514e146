#!/bin/bash
# first contact of the compact MSM pipeline with the GPU: its parity tests, then time vs n
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msm_compact_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 300 python tools/msm_vs_n.py > gpurun_out/r04_msm_vs_n.txt 2>&1; cat gpurun_out/r04_msm_vs_n.txt
