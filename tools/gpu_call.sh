# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  timeout 200 python tools/ab_kernels.py 1024 8 normal 2>&1 | tail -1
  GG_K5_DEBUG=1 timeout 200 python tools/ab_kernels.py 1024 8 k5_no_gathers 2>&1 | tail -1
  GG_K5_DEBUG=2 timeout 200 python tools/ab_kernels.py 1024 8 k5_no_stores 2>&1 | tail -1
  GG_K5_DEBUG=4 timeout 200 python tools/ab_kernels.py 1024 8 k5_no_atomics 2>&1 | tail -1
  GG_K5_DEBUG=7 timeout 200 python tools/ab_kernels.py 1024 8 k5_none_of_them 2>&1 | tail -1
done | tee gpurun_out/t34_ab.log
