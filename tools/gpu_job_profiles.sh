#!/bin/bash
# the round's profile set (GPU box): rocprofv3 kernel traces of the bench command, of the other configs, of the
# unchanged hadamard 20, time vs n, one isolated compact call; everything under gpurun_out/<tag>_*
cd "$(dirname "$0")/.." || exit 1
TAG=${TAG:-r05_v3}
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
sh tools/profile_spread.sh $TAG > /dev/null 2>&1
# the other configs
mkdir -p "$OUT/prof_cfg"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_cfg" -o trace -- python3 bench.py --only-configs g2_msm,cppoly,pairing,cphad_verify,fr_fold,ntt > "$OUT/${TAG}_configs.json" 2> /dev/null
DB=$(find "$OUT/prof_cfg" -name '*.db' | head -1)
python3 tools/rocpd_summary.py trace "$DB" > "$OUT/${TAG}_configs_kernel_trace_stats.txt"
python3 tools/rocpd_summary.py spread "$DB" > "$OUT/${TAG}_configs_kernel_spread.txt"
rm -rf "$OUT/prof_cfg"
# the unchanged reference binary
TAG=$TAG sh tools/run_hadamard.sh 12 16 20 > /dev/null 2>&1
python3 tools/msm_vs_n.py > "$OUT/${TAG}_msm_vs_n.txt" 2>&1
# round 5: G2 MSM vs n on a 2^16-point handle, compact pipeline and the general one; the NTT; first calls of fresh processes;
# the fused Miller kernel's timing-only experiments
python3 tools/msm_vs_n.py g2 16 16 > "$OUT/${TAG}_g2_msm_vs_n.txt" 2>&1
LSA_NO_COMPACT_G2=1 python3 tools/msm_vs_n.py g2 16 16 > "$OUT/${TAG}_g2_msm_vs_n_general_pipeline.txt" 2>&1
python3 tools/bench_fr_vec.py ntt > "$OUT/${TAG}_ntt.txt" 2>&1
python3 tools/bench_fr_vec.py > "$OUT/${TAG}_fr_vec.txt" 2>&1
python3 tools/cold_msm.py --runs 8 --settings "" --settings LSA_H2D=direct --settings LSA_H2D_STREAMS=0 > "$OUT/${TAG}_cold_msm_g1.txt" 2>&1
python3 tools/cold_msm.py --runs 8 --small-pages --settings "" --settings LSA_H2D=direct > "$OUT/${TAG}_cold_msm_g1_small_pages.txt" 2>&1
python3 tools/cold_msm.py --runs 6 --free-temporaries --settings "" --settings LSA_H2D=direct > "$OUT/${TAG}_cold_msm_g1_free_temporaries.txt" 2>&1
python3 tools/cold_msm.py --runs 5 --group g2 --settings "" --settings LSA_G2_PREPARE_OLD=1 --settings LSA_WARM_MB=192 > "$OUT/${TAG}_cold_msm_g2.txt" 2>&1
LSA_TRACE=2 build/reference_cmake/src/examples/hadamard 20 2>&1 >/dev/null | grep "^\[lsa\]" > "$OUT/${TAG}_hadamard_d20_call_trace.txt"
LSA_TRACE=2 LSA_WARM_MB=192 build/reference_cmake/src/examples/hadamard 20 2>&1 >/dev/null | grep -E "grow|msm_g[12] +n=(1048576|524288)" | head -30 > "$OUT/${TAG}_hadamard_d20_growth_with_the_round4_warmup.txt"
for e in 0 1 2 4 3 5 6 7; do LSA_FUSED_EXPERIMENT=$e python3 tools/fused_experiment.py 2>/dev/null | tail -1; done > "$OUT/${TAG}_fused_miller_experiments.txt"
sh tools/profile_cmd_timeline.sh ${TAG}_ntt 12 tools/bench_fr_vec.py ntt > /dev/null 2>&1
sh tools/profile_cmd_timeline.sh ${TAG}_compact_n4096 12 tools/single_call_trace.py 4096 > /dev/null 2>&1
sh tools/profile_cmd_timeline.sh ${TAG}_single_call 30 tools/single_call_trace.py > /dev/null 2>&1
python3 tools/final_exp_probe.py > "$OUT/${TAG}_final_exp.txt" 2>&1
python3 tools/pairing_check_probe.py > "$OUT/${TAG}_pairing_check.txt" 2>&1
sh tools/profile_cmd_timeline.sh ${TAG}_pairing_check 9 tools/pairing_check_probe.py > /dev/null 2>&1
ls -la "$OUT" | grep "$TAG" | awk '{print $5, $9}'
