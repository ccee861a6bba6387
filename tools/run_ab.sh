mkdir -p gpurun_out/r03p; O=gpurun_out/r03p
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/pytest_all.log; grep -E "passed|failed|Error|error" $O/pytest_all.log
GG_HOST_TIMING=1 python tools/host_path_rate.py 2>&1 | tail -4
