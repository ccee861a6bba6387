# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/pair_timing.py 2>&1 | tee gpurun_out/pair_timing.log
timeout 600 python tools/pair_ab.py 1,8 2>&1 | tee gpurun_out/pair_ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/tests.log
