# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04c_tests.log
cat gpurun_out/r04c_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh r04c > gpurun_out/r04c_profile.log 2>&1
tail -3 gpurun_out/r04c_profile.log
bash tools/sq_pmc.sh > gpurun_out/r04c/sq_counters.txt 2>&1
rm -rf gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3
cp profiles/pmc_summary.json gpurun_out/r04c/pmc_summary_final.json
cp gpurun_out/r04c_tests.log gpurun_out/r04c/gpu_tests.log
ls gpurun_out/r04c
