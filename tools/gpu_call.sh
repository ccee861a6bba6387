# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or each_launch_geometry or front_end or reduce or fuzz or dense_single or edge_cases" 2>&1 | tail -8 > gpurun_out/t5_tests.log
cat gpurun_out/t5_tests.log
{
GG_K2_LIGHT_MAX=512 SKIP_SMALL=1 BATCHES_BIG=1,128 timeout 600 python tools/latency_probe.py 2>/dev/null | tail -1
SKIP_SMALL=1 BATCHES_BIG=1,128 timeout 600 python tools/latency_probe.py 2>/dev/null | tail -1
GG_PW=16384 SKIP_SMALL=1 BATCHES_BIG=128 timeout 600 python tools/latency_probe.py 2>/dev/null | tail -1
GG_K2_LIGHT_MAX=1024 timeout 200 python tools/ab_kernels.py 1024 8 light1024 2>/dev/null | tail -1
timeout 200 python tools/ab_kernels.py 1024 8 light512 2>/dev/null | tail -1
} | tee gpurun_out/t5_ab.log
