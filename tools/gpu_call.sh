# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/t42_tests.log
V=$GRAFT_REPO_ROOT/groundgrid_amd/variants
for rep in 1 2 3; do
  GROUNDGRID_HIP_LIB=$V/lib_base.so timeout 200 python tools/ab_kernels.py 1024 8 base 2>&1 | tail -1
  timeout 200 python tools/ab_kernels.py 1024 8 k3_fused_3x3_sums 2>&1 | tail -1
done | tee gpurun_out/t42_ab.log
GROUNDGRID_HIP_LIB=$V/lib_base.so timeout 300 python tools/ab_config4.py 128 base 2>&1 | tail -1 | tee gpurun_out/t42_c4.log
timeout 300 python tools/ab_config4.py 128 k3_fused_3x3_sums 2>&1 | tail -1 | tee -a gpurun_out/t42_c4.log
