# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stages_wire.py -m gpu -x -q -k "results_straight or fused or graph" > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/t.log | tail -5
