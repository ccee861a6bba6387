cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages_wire.py -m gpu -x -q -k "synthetic_drive or unfenced" 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -8 | tee gpurun_out/tests_new.log
timeout 1500 python bench.py > gpurun_out/bench_r06a.json 2> gpurun_out/bench_r06a.err; tail -c 3000 gpurun_out/bench_r06a.json; tail -5 gpurun_out/bench_r06a.err
