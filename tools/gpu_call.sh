cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/pair_timing.py 2>&1 | grep "wg 1" | tee gpurun_out/pair_timing.log
timeout 900 python tools/pair_ab.py 1 2>&1 | grep -v amdgpu | tee gpurun_out/pair_ab.log
timeout 900 python tools/pair_race.py 6 6 2>&1 | grep -v amdgpu.ids | cut -c1-300 | sort | uniq -c
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -3 | tee gpurun_out/tests.log
