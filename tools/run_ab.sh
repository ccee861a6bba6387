mkdir -p gpurun_out/r03n; O=gpurun_out/r03n
V=$PWD/groundgrid_amd/variants
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sweep or geometr or decay or hdl64 or largest" 2>&1 | tail -12) > $O/pytest.log; grep -E "passed|failed|Error|error" $O/pytest.log
for i in 1 2; do
GROUNDGRID_HIP_LIB=$V/lib_head.so timeout 300 python tools/ab_kernels.py 1024 8 head > $O/ab_head$i.json 2>>$O/err.log; cat $O/ab_head$i.json
timeout 300 python tools/ab_kernels.py 1024 8 new > $O/ab_new$i.json 2>>$O/err.log; cat $O/ab_new$i.json
done
SKIP_BIG=1 BATCHES_SMALL=1 timeout 300 python tools/latency_probe.py > $O/lat.json 2>>$O/err.log; cat $O/lat.json
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/pytest_all.log; grep -E "passed|failed|Error|error" $O/pytest_all.log
