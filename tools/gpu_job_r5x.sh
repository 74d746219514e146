#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 1500 python -m pytest tests/test_fr_vec_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -4
