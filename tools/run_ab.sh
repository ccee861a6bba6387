mkdir -p gpurun_out/r03l; O=gpurun_out/r03l
V=$PWD/groundgrid_amd/variants
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/pytest.log; grep -E "passed|failed|Error|error" $O/pytest.log
GROUNDGRID_HIP_LIB=$V/lib_head.so python tools/ab_kernels.py 1024 8 head > $O/ab_head.json 2>>$O/err.log; cat $O/ab_head.json
for d in 0 8 12 16 24 32; do GG_K2_DENSE_WGS=$d python tools/ab_kernels.py 1024 8 dense$d > $O/ab_d$d.json 2>>$O/err.log; cat $O/ab_d$d.json; done
GROUNDGRID_HIP_LIB=$V/lib_head.so python tools/ab_kernels.py 1024 8 head > $O/ab_head2.json 2>>$O/err.log; cat $O/ab_head2.json
