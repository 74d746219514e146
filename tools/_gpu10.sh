cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export LSA_CRS_TABLE_AFTER=0
build/h2d_vectors x 4096 | tail -3
mkdir -p gpurun_out/prof_p
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_p -o trace -- build/h2d_vectors x 4096 > /dev/null 2>&1
DB=$(find gpurun_out/prof_p -name '*.db' | head -1)
python3 tools/rocpd_summary.py spread "$DB" | head -30
rm -rf gpurun_out/prof_p
