cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in 64 4096 65536; do
  mkdir -p gpurun_out/prof_s$n
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s$n -o trace -- python3 tools/small_msm_trace.py $n > /dev/null 2>&1
  DB=$(find gpurun_out/prof_s$n -name '*.db' | head -1)
  echo "== n=$n"
  python3 tools/rocpd_summary.py spread "$DB" | head -24
  rm -rf gpurun_out/prof_s$n
done
