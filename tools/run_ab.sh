mkdir -p gpurun_out/r03m; O=gpurun_out/r03m
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log; grep -E "passed|failed|Error|error|assert" $O/pytest.log | head -20
