#!/bin/bash
# the unchanged hadamard example several times per setting: the msm_g1 / pairing lines of the shim's statistics
cd "$(dirname "$0")/.." || exit 1
d=${1:-20}
for setting in "LSA_CRS_PREFIX_TABLE=65536" "LSA_CRS_PREFIX_TABLE=16384" "LSA_CRS_PREFIX_TABLE=0"; do
  for rep in 1 2 3; do
    env $setting LSA_SHIM_STATS=1 build/reference/hadamard $d 2>&1 >/dev/null | grep lsa_shim_stats | python3 -c "
import json,sys
st=json.loads(sys.stdin.read())['lsa_shim_stats']
print('$setting', 'msm_g1 %.1f ms' % st['msm_g1']['ms'], 'pairing %.1f' % st['pairing']['ms'], 'prepare %.1f' % st['msm_host_path']['bases_prepare_ms'], 'fp_wait %.1f' % st['msm_host_path']['fingerprint_wait_ms'], 'kernels %.1f' % st['msm_host_path']['kernels_ms'], 'h2d %.1f' % st['msm_host_path']['h2d_scalars_ms'])"
  done
done
