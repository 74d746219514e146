cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -5
for m in direct auto; do echo $m; LSA_H2D=$m build/h2d_vectors 2>&1 | grep "rep"; done
export LSA_SHIM_STATS=1
build/reference/hadamard 20 2>&1 | grep "TOTAL\|CPPoly\|lipmaa\|msm_host_path" | sed "s/.*msm_host_path/msm_host_path/"
