mkdir -p gpurun_out/r03i; O=gpurun_out/r03i
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/pytest.log; cat $O/pytest.log
GG_HOST_TIMING=1 python tools/host_path_rate.py > $O/host_path.txt 2>&1; cat $O/host_path.txt | tail -6
python tools/ab_kernels.py 1024 8 k3dead > $O/ab.json 2>>$O/err.log; cat $O/ab.json
