cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/pair_race.py 8 8 2>&1 | grep -v amdgpu.ids | cut -c1-600 | sort | uniq -c | tee gpurun_out/pair_race.log
timeout 300 python tools/pair_timing.py 2>&1 | grep "wg 1" | tee gpurun_out/pair_timing.log
timeout 600 python tools/pair_ab.py 1 2>&1 | tee gpurun_out/pair_ab.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt1 -o kt -- python $GRAFT_REPO_ROOT/tools/k4_run.py 1 0 60 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/kt1 -name '*kernel_stats.csv' | head -1) gpurun_out/pair_kernel_stats_b1.csv; rm -rf gpurun_out/kt1
grep "k_sweep" gpurun_out/pair_kernel_stats_b1.csv | cut -d, -f1-4 | cut -c1-40,100-200
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/tests.log
