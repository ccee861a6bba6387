# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  timeout 200 python tools/ab_kernels.py 1024 8 normal 2>&1 | tail -1
  GG_K2_DEBUG=8 timeout 200 python tools/ab_kernels.py 1024 8 without_the_64_fullest_cells_chains 2>&1 | tail -1
  GG_K2_DEBUG=3 timeout 200 python tools/ab_kernels.py 1024 8 dense_tiles_stop_after_placing 2>&1 | tail -1
  GG_K2_SKIP=2 timeout 200 python tools/ab_kernels.py 1024 8 no_dense_tiles 2>&1 | tail -1
  GG_K2_SKIP=1 timeout 200 python tools/ab_kernels.py 1024 8 no_light_tiles 2>&1 | tail -1
done | tee gpurun_out/t29_ab.log
