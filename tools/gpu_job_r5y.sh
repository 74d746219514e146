#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 1500 python -m pytest tests/test_fr_vec_gpu.py tests/test_full_size_gpu.py -x -q -k ntt 2>&1 | tail -3
python tools/bench_fr_vec.py ntt 2>/dev/null | grep -E '"log_n": (16|20|22|24)' | cut -c1-200
