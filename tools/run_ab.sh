mkdir -p gpurun_out/r03r; O=gpurun_out/r03r
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/pytest_all.log; grep -E "passed|failed|Error|error" $O/pytest_all.log
for t in 1 2 4 6; do echo threads $t; GG_HOST_THREADS=$t GG_HOST_TIMING=1 python tools/host_path_rate.py 2>&1 | tail -3; done
