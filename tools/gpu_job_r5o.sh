#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
LSA_TRACE=2 build/reference_cmake/src/examples/hadamard 20 2>&1 >/dev/null | grep -E "grow|msm_g2|msm_g1 +n=(1048576|524288)|fr_ntt|batch_exp" | head -60
LSA_TRACE=2 build/reference_cmake/src/examples/cplink 2>&1 >/dev/null | grep -E "grow" | head
