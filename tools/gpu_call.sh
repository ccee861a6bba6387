# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/t7_tests.log
cat gpurun_out/t7_tests.log
{
for rep in 1 2; do
GG_K3=1 timeout 200 python tools/ab_kernels.py 1024 8 k3_bands 2>/dev/null | tail -1
timeout 200 python tools/ab_kernels.py 1024 8 k3_tiles 2>/dev/null | tail -1
done
GG_K3=1 SKIP_BIG=1 BATCHES_SMALL=1 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
SKIP_BIG=1 BATCHES_SMALL=1 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
GG_K3=1 SKIP_SMALL=1 BATCHES_BIG=128 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
SKIP_SMALL=1 BATCHES_BIG=128 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1
} | tee gpurun_out/t7_ab.log
