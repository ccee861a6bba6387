mkdir -p gpurun_out/r03s; O=gpurun_out/r03s
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "sweep_cut" 2>&1 | tail -12) > $O/pytest_sweep.log; grep -E "passed|failed|Error|error" $O/pytest_sweep.log
for sp in 2 0; do GG_SWEEP_SPLIT=$sp BATCHES_SMALL=1,8,64 BATCHES_BIG=1,8 timeout 300 python tools/latency_probe.py > $O/lat_split$sp.json 2>>$O/err.log; cat $O/lat_split$sp.json; done
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/pytest_all.log; grep -E "passed|failed|Error|error" $O/pytest_all.log
