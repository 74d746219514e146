#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for r in 1 2 3; do echo "rounds per launch: $r"; LSA_FOLD_ROUNDS=$r LSA_FOLD_ROUNDS_HALVES=$r python tools/bench_fr_vec.py 2>/dev/null | grep '"d": 24' | cut -c1-150; done
