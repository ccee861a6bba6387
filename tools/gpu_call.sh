# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/t11_tests.log
cat gpurun_out/t11_tests.log
V=$PWD/groundgrid_amd/variants
{
for rep in 1 2 3; do
GROUNDGRID_HIP_LIB=$V/lib_base.so timeout 200 python tools/ab_kernels.py 1024 8 base 2>/dev/null | tail -1
timeout 200 python tools/ab_kernels.py 1024 8 k1_no_gather_cold 2>/dev/null | tail -1
done
} | tee gpurun_out/t11_ab.log
