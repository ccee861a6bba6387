cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
F='RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|amdgpu'
timeout 2400 python tools/fuzz_fresh.py 400 2400 2>&1 | grep -v "$F" > gpurun_out/r06_fuzz_fresh_2.log; tail -2 gpurun_out/r06_fuzz_fresh_2.log
timeout 2400 python tools/fuzz_knobs.py 400 2000 2>&1 | grep -v "$F" > gpurun_out/r06_fuzz_knobs_2.log; tail -2 gpurun_out/r06_fuzz_knobs_2.log
timeout 2400 python tools/fuzz_more.py 600 3000 2>&1 | grep -v "$F" > gpurun_out/r06_fuzz_more_2.log; tail -2 gpurun_out/r06_fuzz_more_2.log
timeout 1200 python tools/fuzz_walk.py 0 300 2>&1 | grep -v "$F" > gpurun_out/r06_fuzz_walk.log; tail -2 gpurun_out/r06_fuzz_walk.log
