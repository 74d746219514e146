#!/bin/bash
cd /root/repo
python -m pytest tests/test_fr_vec_gpu.py tests/test_full_size_gpu.py tests/test_host_copies_gpu.py -x -q -m gpu -k "not ntt and not step" 2>&1 | tail -3
python tools/bench_fr_vec.py 2>&1 | grep -v oracle | tail -8
