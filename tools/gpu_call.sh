# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/t10_tests.log
cat gpurun_out/t10_tests.log
V=$PWD/groundgrid_amd/variants
{
for rep in 1 2 3; do
GROUNDGRID_HIP_LIB=$V/lib_base.so timeout 200 python tools/ab_kernels.py 1024 8 base 2>/dev/null | tail -1
timeout 200 python tools/ab_kernels.py 1024 8 scatter_prefetch 2>/dev/null | tail -1
done
GROUNDGRID_HIP_LIB=$V/lib_base.so SKIP_SMALL=1 BATCHES_BIG=128 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
SKIP_SMALL=1 BATCHES_BIG=128 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
} | tee gpurun_out/t10_ab.log
