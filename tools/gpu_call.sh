cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -6 | tee gpurun_out/tests.log
timeout 1500 python bench.py > gpurun_out/bench_r06b.json 2> gpurun_out/bench_r06b.err; tail -c 2600 gpurun_out/bench_r06b.json; tail -3 gpurun_out/bench_r06b.err
