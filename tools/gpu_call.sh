# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/direct_probe.py 2>&1 | tail -1 | tee gpurun_out/direct_probe.json
timeout 900 python -m pytest tests -m gpu -x -q -k "not full_size" > gpurun_out/tests_direct.log 2>&1; grep -E "passed|failed|error" gpurun_out/tests_direct.log | tail -3
