# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  timeout 200 python tools/ab_kernels.py 1024 8 normal 2>&1 | tail -1
  GG_K3_DEBUG=3 timeout 200 python tools/ab_kernels.py 1024 8 k3_no_cell_passes 2>&1 | tail -1
  GG_K3_DEBUG=1 timeout 200 python tools/ab_kernels.py 1024 8 k3_prologue_only 2>&1 | tail -1
done | tee gpurun_out/t33_ab.log
