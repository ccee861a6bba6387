mkdir -p gpurun_out/r03k; O=gpurun_out/r03k
V=$PWD/groundgrid_amd/variants
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/pytest.log; grep -E "passed|failed|Error|error" $O/pytest.log
for i in 1 2; do
GROUNDGRID_HIP_LIB=$V/lib_head.so python tools/ab_kernels.py 1024 8 head > $O/ab_head$i.json 2>>$O/err.log; cat $O/ab_head$i.json
python tools/ab_kernels.py 1024 8 new > $O/ab_new$i.json 2>>$O/err.log; cat $O/ab_new$i.json
done
