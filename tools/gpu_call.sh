cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/pair_race.py 12 30 2>&1 | grep -v amdgpu.ids | cut -c1-600 | sort | uniq -c | tee gpurun_out/pair_race.log
timeout 300 python tools/pair_timing.py 2>&1 | grep "wg 1" | tee gpurun_out/pair_timing.log
timeout 600 python tools/pair_ab.py 1 2>&1 | tee gpurun_out/pair_ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/tests.log
