# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  timeout 200 python tools/ab_kernels.py 1024 8 normal 2>&1 | tail -1
  GG_K2_DEBUG=4 timeout 200 python tools/ab_kernels.py 1024 8 chains_read_one_line_per_lane 2>&1 | tail -1
  GG_K2_DEBUG=7 timeout 200 python tools/ab_kernels.py 1024 8 chains_read_coalesced 2>&1 | tail -1
done | tee gpurun_out/t24_ab.log
