# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05e/gpu_tests_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/r05e/gpu_tests_full.log | tail -2 | tee gpurun_out/r05e/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r05e > gpurun_out/r05e/profile_round.log 2>&1
tail -c 600 gpurun_out/r05e/bench_default.json
timeout 100 python tools/host_call_probe.py 2>&1 | tail -1 | tee gpurun_out/r05e/host_call_probe.json
