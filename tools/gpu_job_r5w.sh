#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 900 python -m pytest tests/test_fr_vec_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -2
python tools/bench_fr_vec.py 2>/dev/null | cut -c1-200
LSA_FR_GRAPHS=0 python tools/bench_fr_vec.py 2>/dev/null | grep '"d": 24' | cut -c1-200
timeout 900 python tools/bench_configs.py --only fr_fold,cppoly 2>/dev/null | cut -c1-600
