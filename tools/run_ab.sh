mkdir -p gpurun_out/r03e; O=gpurun_out/r03e
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/pytest.log; cat $O/pytest.log
V=$PWD/groundgrid_amd/variants
python tools/ab_kernels.py 1024 8 zcap2944 > $O/ab_base.json 2>$O/err.log; cat $O/ab_base.json
GG_K2_GLOBAL_PATH=1 python tools/ab_kernels.py 1024 8 globalpath > $O/ab_global.json 2>>$O/err.log; cat $O/ab_global.json
for z in 3200 4096 4864; do GROUNDGRID_HIP_LIB=$V/lib_zcap$z.so python tools/ab_kernels.py 1024 8 zcap$z > $O/ab_z$z.json 2>>$O/err.log; cat $O/ab_z$z.json; done
python tools/k2_phases.py > $O/k2_phases.txt 2>>$O/err.log; cat $O/k2_phases.txt
tail -3 $O/err.log
