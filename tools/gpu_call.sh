# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t4_tests.log
cat gpurun_out/t4_tests.log
V=$PWD/groundgrid_amd/variants
{
GROUNDGRID_HIP_LIB=$V/lib_r03.so SKIP_BIG=1 BATCHES_SMALL=1,8 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
SKIP_BIG=1 BATCHES_SMALL=1,8 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
GG_FRONT=3 SKIP_BIG=1 BATCHES_SMALL=1,8 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
GROUNDGRID_HIP_LIB=$V/lib_r03.so SKIP_SMALL=1 BATCHES_BIG=1 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
SKIP_SMALL=1 BATCHES_BIG=1 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
for rep in 1 2; do
  GROUNDGRID_HIP_LIB=$V/lib_r03.so timeout 200 python tools/ab_kernels.py 1024 8 r03 2>/dev/null | tail -1
  timeout 200 python tools/ab_kernels.py 1024 8 new 2>/dev/null | tail -1
done
} | tee gpurun_out/t4_ab.log
timeout 900 python bench.py > gpurun_out/t4_bench.json 2> gpurun_out/t4_bench.err
tail -c 3000 gpurun_out/t4_bench.json
