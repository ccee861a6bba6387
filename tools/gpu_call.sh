# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r05d_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r05d_gpu_tests.log
bash tools/profile_round.sh r05d > gpurun_out/r05d_profile_round.log 2>&1
tail -3 gpurun_out/r05d_profile_round.log | cut -c1-400
